set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_infer.py tests/test_golden.py tests/test_gpu_infer_main.py tests/test_gpu_trained_precision.py -x -q > gpurun_out/call17_tests.txt 2>&1 || true
tail -5 gpurun_out/call17_tests.txt
for r in 1 2 3 4; do for v in 0 1; do
  i=$(ANH_WS_INFER_PSTORE=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "INFER_PSTORE=$v infer $i"
done; done 2>&1 | tee gpurun_out/call17_ab.txt
