set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
bash tools/ab_env.sh 3 "ANH_LIBRARY=$P/annonet_amd/lib/libannonet_hip.so" "ANH_LIBRARY=$P/annonet_amd/lib_nt_LOADS/libannonet_hip.so" "ANH_LIBRARY=$P/annonet_amd/lib_nt_STORES/libannonet_hip.so" "ANH_LIBRARY=$P/annonet_amd/lib_nt_BOTH/libannonet_hip.so" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call9_ab.txt
