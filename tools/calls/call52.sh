set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
ANH_WGRAD_XCD_BANDS=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_first_step_dirty_memory.py -x -q > gpurun_out/call52_tests.txt 2>&1 || true
tail -3 gpurun_out/call52_tests.txt
grep -q "failed\|rror" gpurun_out/call52_tests.txt && exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_schedules.py -x -q 2>&1 | tail -2
AB_NO_INFER=1 bash tools/ab_env.sh 5 "ANH_WGRAD_XCD_BANDS=0" "ANH_WGRAD_XCD_BANDS=1" 2>&1 | tee gpurun_out/call52_ab.txt
