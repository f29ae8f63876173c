set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_infer.py tests/test_golden.py tests/test_gpu_schedules.py -x -q > gpurun_out/call7_tests.txt 2>&1 || true
tail -5 gpurun_out/call7_tests.txt
for r in 1 2 3; do for v in 0 1; do
  i=$(ANH_WS_DEFER_STORES=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "DEFER_STORES=$v infer $i"
done; done 2>&1 | tee gpurun_out/call7_ab.txt
AB_NO_INFER=1 bash tools/ab_env.sh 3 "ANH_WS_PSTAT=1" "ANH_WS_PSTAT=5" "ANH_WS_PSTAT=6" 2>&1 | tee -a gpurun_out/call7_ab.txt
for v in 1 5; do
  ANH_WS_PSTAT=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" | grep dgrad | awk -v v=$v '{print "PSTAT="v, $1, $2}'
done
