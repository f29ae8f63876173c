set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
# filter registers compiled out (lib_prev) against the default build, and against the default build with them switched off at run time
bash tools/ab_env.sh 4 "ANH_LIBRARY=$P/annonet_amd/lib_prev/libannonet_hip.so" "-" "ANH_WS_FILTER_REGS=0" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call38_ab.txt
