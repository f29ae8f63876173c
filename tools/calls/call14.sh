set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_schedules.py -x -q -k "producer_waves or default" > gpurun_out/call14_tests.txt 2>&1 || true
tail -5 gpurun_out/call14_tests.txt
AB_NO_INFER=1 bash tools/ab_env.sh 3 "ANH_WS_PSTAT=1" "ANH_WS_PSTAT=7" "ANH_WS_PSTAT=8" 2>&1 | tee gpurun_out/call14_ab.txt
for v in 1 7; do
  ANH_WS_PSTAT=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" | grep dgrad | awk -v v=$v '{print "PSTAT="v, $1, $2}'
done
