set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
bash tools/ab_env.sh 3 "ANH_LIBRARY=$P/annonet_amd/lib/libannonet_hip.so" "ANH_LIBRARY=$P/annonet_amd/lib_pin/libannonet_hip.so" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call20_ab.txt
ANH_LIBRARY=$P/annonet_amd/lib_pin/libannonet_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_golden.py -x -q 2>&1 | tail -3
