set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
bash tools/ab_env.sh 5 "ANH_LIBRARY=$P/annonet_amd/lib_prev/libannonet_hip.so" "-" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call49_ab.txt
