set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
for r in 1 2 3; do
  for n in lib_prev lib; do
    i=$(ANH_LIBRARY=$P/annonet_amd/$n/libannonet_hip.so python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "$n infer $i"
  done
done 2>&1 | tee gpurun_out/call2_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_first_step_dirty_memory.py tests/test_gpu_parity.py tests/test_gpu_sharded_infer.py tests/test_golden.py -x -q > gpurun_out/call2_tests.txt 2>&1 || true
tail -5 gpurun_out/call2_tests.txt
cd /tmp && export TMPDIR=/tmp && cd $P
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itl -- python3 bench.py --mode infer --steps 4 --warmup 1 --no-cpu-baseline --prewarm-s 0 > gpurun_out/call2_itl.json 2> gpurun_out/call2_itl.err
python3 tools/timeline.py gpurun_out/itl/*/*_kernel_trace.csv argmax_kernel > gpurun_out/call2_infer_timeline.txt 2>&1 || true
head -60 gpurun_out/call2_infer_timeline.txt
# the bench with the new CPU baseline (timing of the default line)
( time python bench.py > gpurun_out/call2_bench.json 2> gpurun_out/call2_bench.err ) 2>&1 | tail -3
rm -rf gpurun_out/itl
