set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_infer.py tests/test_gpu_parity.py tests/test_golden.py -x -q > gpurun_out/call19_tests.txt 2>&1 || true
tail -8 gpurun_out/call19_tests.txt
for r in 1 2 3; do for v in 0 1 2; do
  i=$(ANH_WS_DMA=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "DMA=$v infer $i"
done; done 2>&1 | tee gpurun_out/call19_ab.txt
