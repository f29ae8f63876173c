set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_sharded_infer.py tests/test_gpu_infer_main.py tests/test_gpu_ops.py -x -q > gpurun_out/call46_tests.txt 2>&1 || true
tail -4 gpurun_out/call46_tests.txt
grep -q "failed\|rror" gpurun_out/call46_tests.txt && exit 1
for r in 1 2 3 4; do for v in 0 1; do
  i=$(ANH_WS_TALL=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "[ANH_WS_TALL=$v] $i"
done; done 2>&1 | tee gpurun_out/call46_ab.txt
for v in 0 1; do ANH_WS_TALL=$v ANH_BENCH_VERBOSE=1 python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --prewarm-s 1 2>&1 | grep "ms/image" | head -9 > gpurun_out/call46_v$v.txt; done
paste gpurun_out/call46_v0.txt gpurun_out/call46_v1.txt | awk '{print $1, $2, $8}'
