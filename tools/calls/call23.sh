set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
ANH_WS_TEAMS=1 timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_first_step_dirty_memory.py -x -q > gpurun_out/call23_tests_teams.txt 2>&1 || true
tail -6 gpurun_out/call23_tests_teams.txt
grep -q "failed\|error" gpurun_out/call23_tests_teams.txt && exit 1
bash tools/ab_env.sh 3 "ANH_WS_TEAMS=0" "ANH_WS_TEAMS=1" "ANH_WS_TEAMS=2" "ANH_WS_TEAMS=3" 2>&1 | tee gpurun_out/call23_ab.txt
