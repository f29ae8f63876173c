set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_first_step_dirty_memory.py tests/test_gpu_schedules.py -x -q > gpurun_out/call40_tests.txt 2>&1 || true
tail -4 gpurun_out/call40_tests.txt
grep -q "failed\|rror" gpurun_out/call40_tests.txt && exit 1
AB_NO_INFER=1 bash tools/ab_env.sh 5 "ANH_LIBRARY=$P/annonet_amd/lib_prev/libannonet_hip.so" "-" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call40_ab.txt
