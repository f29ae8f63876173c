set -e
out=gpurun_out/insts_infer
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $out
I="--mode infer --no-cpu-baseline --prewarm-s 0"
python3 bench.py $I --steps 2 --warmup 1 --dump-launch-order $out/launch_order.json > /dev/null 2> $out/order.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq -- python3 bench.py $I --steps 3 --warmup 1 > /dev/null 2> $out/sq.err || (tail -5 $out/sq.err; exit 1)
python3 tools/pmc_counters.py $out/launch_order.json $out/sq/*/*_counter_collection.csv $out/insts.json > $out/insts.txt
rm -rf $out/sq
cut -c1-230 $out/insts.txt
