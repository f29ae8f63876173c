set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -2
AB_NO_INFER=1 bash tools/ab_env.sh 6 "ANH_LIBRARY=$P/annonet_amd/lib_prev/libannonet_hip.so" "-" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call56_ab.txt
