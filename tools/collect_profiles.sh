#!/bin/bash
# Collects the round's measurement set on ONE MI355X box into gpurun_out/prof (copy what is judged into profiles/ with
# tools/publish_profiles.py): the bench line, rocprofv3 kernel stats + trace of the same command, the two HBM counter passes, one SQ
# counter pass, and the inference lines.  Counter passes run on their own (never with a trace domain other than --kernel-trace).
#   usage (on the GPU box): bash tools/collect_profiles.sh
set -e
out=gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
# launch orders of one step / one image (the counter passes below are mapped to profiler entries through them)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --prewarm-s 0 --dump-launch-order $out/launch_order.json > /dev/null 2> $out/order.err
# profiled runs: the training process alone (--no-infer; --prewarm-s 0 keeps the traces short)
P="--no-cpu-baseline --no-infer --prewarm-s 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 100 --warmup 10 $P > $out/bench_under_rocprof.json 2> $out/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 30 --warmup 5 $P > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 30 --warmup 5 $P > /dev/null 2> $out/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq -- python3 bench.py --steps 20 --warmup 5 $P > /dev/null 2> $out/sq.err
# tiled inference (BASELINE.json configs[2]): the line, the kernel stats of the same command, the two HBM counter passes
I="--mode infer --no-cpu-baseline --prewarm-s 0"
python3 bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --prewarm-s 0 --dump-launch-order $out/infer_launch_order.json > /dev/null 2> $out/infer.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ikt -- python3 bench.py $I --steps 20 --warmup 3 > $out/bench_infer_under_rocprof.json 2> $out/ikt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/ifetch -- python3 bench.py $I --steps 6 --warmup 2 > /dev/null 2> $out/ifetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/iwrite -- python3 bench.py $I --steps 6 --warmup 2 > /dev/null 2> $out/iwrite.err
python3 bench.py --mode infer --precision fp32 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_infer_4096_fp32.json 2>> $out/infer.err
python3 bench.py --mode infer --image-side 16384 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_infer_16384_bf16.json 2>> $out/infer.err
ANH_CONCURRENT_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $out/kt1 -- python3 bench.py --steps 40 --warmup 10 $P > $out/bench_one_stream_under_rocprof.json 2> $out/kt1.err
ANH_CONCURRENT_WGRAD=0 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 > $out/bench_one_stream.json 2> $out/one.err
# stand-alone times of every layer x pass (one stream, no concurrent filter gradients): the MFMA-bound layers against the peak
python3 tools/standalone_table.py $out/bench_one_stream.json > $out/standalone_table.txt
# the default line (training + the `infer` object measured by its child process), as the driver runs it — LAST, so that the
# `traffic` fields of its roofline objects come from the counter passes of THIS collection
python3 tools/pmc_traffic.py $out/launch_order.json $out/fetch/*/*_counter_collection.csv $out/write/*/*_counter_collection.csv $out/traffic.json > $out/traffic.txt
python3 tools/pmc_traffic.py $out/infer_launch_order.json $out/ifetch/*/*_counter_collection.csv $out/iwrite/*/*_counter_collection.csv $out/infer_traffic.json > $out/infer_traffic.txt
python3 tools/pmc_counters.py $out/launch_order.json $out/sq/*/*_counter_collection.csv $out/sq_counters.json > $out/sq_counters.txt
export ANH_TRAFFIC_JSON=$PWD/$out/traffic.json ANH_INFER_TRAFFIC_JSON=$PWD/$out/infer_traffic.json ANH_SQ_JSON=$PWD/$out/sq_counters.json
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --mode infer > $out/bench_infer_4096_bf16.json 2>> $out/infer.err
echo collected; du -sh $out
