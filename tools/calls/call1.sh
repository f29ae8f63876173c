set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash tools/ab_env.sh 3 "ANH_WS_XCD_BANDS=0" "ANH_WS_XCD_BANDS=1" 2>&1 | tee gpurun_out/call1_ab.txt
for v in 0 1; do
  ANH_WS_XCD_BANDS=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 > gpurun_out/call1_verbose_$v.txt 2>&1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/call1_tests.txt 2>&1; tail -5 gpurun_out/call1_tests.txt
