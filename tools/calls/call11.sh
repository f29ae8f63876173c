set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_sharded_infer.py tests/test_gpu_first_step_dirty_memory.py -x -q > gpurun_out/call11_tests.txt 2>&1 || true
tail -5 gpurun_out/call11_tests.txt
bash tools/ab_env.sh 3 "ANH_WS_TSTORE=0" "ANH_WS_TSTORE=1" 2>&1 | tee gpurun_out/call11_ab.txt
for v in 0 1; do
  ANH_WS_TSTORE=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" | awk '{print $1, $2}' | sort > gpurun_out/call11_v$v.txt
done
paste gpurun_out/call11_v0.txt gpurun_out/call11_v1.txt | awk '{printf "%-48s %s %s %+.0f\n", $1, $2, $4, ($4-$2)*1000}' | grep "conv_mfma"
