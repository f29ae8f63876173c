set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_sharded_infer.py tests/test_gpu_multidev.py tests/test_gpu_infer_main.py tests/test_gpu_det_seed_order.py tests/test_gpu_exit_order.py -x -q > gpurun_out/call42_tests.txt 2>&1 || true
tail -4 gpurun_out/call42_tests.txt
grep -q "failed\|rror" gpurun_out/call42_tests.txt && exit 1
for r in 1 2 3 4; do for v in 0 1; do
  i=$(ANH_INFER_ASYNC_BLEND=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*\|"value_labels_on_host": [0-9.]*' | head -2 | tr '\n' ' ')
  echo "[ANH_INFER_ASYNC_BLEND=$v] $i"
done; done 2>&1 | tee gpurun_out/call42_ab.txt
ANH_INFER_ASYNC_BLEND=1 python bench.py --mode infer --image-side 16384 --steps 3 --warmup 1 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | tee -a gpurun_out/call42_ab.txt
ANH_INFER_ASYNC_BLEND=0 python bench.py --mode infer --image-side 16384 --steps 3 --warmup 1 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | tee -a gpurun_out/call42_ab.txt
