set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_first_step_dirty_memory.py tests/test_gpu_sharded_infer.py -x -q > gpurun_out/call32_tests.txt 2>&1 || true
tail -6 gpurun_out/call32_tests.txt
grep -q "failed\|rror" gpurun_out/call32_tests.txt && exit 1
bash tools/ab_env.sh 4 "ANH_WS_FILTER_REGS=0" "ANH_WS_FILTER_REGS=1" 2>&1 | tee gpurun_out/call32_ab.txt
