set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_sharded_infer.py tests/test_gpu_infer_main.py -x -q > gpurun_out/call44_tests.txt 2>&1 || true
tail -4 gpurun_out/call44_tests.txt
grep -q "failed\|rror" gpurun_out/call44_tests.txt && exit 1
for r in 1 2 3 4; do for lib in lib_prev lib; do
  i=$(ANH_LIBRARY=$P/annonet_amd/$lib/libannonet_hip.so python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "[$lib] $i"
done; done 2>&1 | tee gpurun_out/call44_ab.txt
