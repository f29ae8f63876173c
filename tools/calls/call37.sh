set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
# the final build against itself with the round's four kept changes switched off (band walk, producer-issued stores, buffer loads, filter registers)
bash tools/ab_env.sh 4 "ANH_LIBRARY=$P/annonet_amd/lib_prev/libannonet_hip.so ANH_WS_XCD_BANDS=0 ANH_WS_PSTAT=1 ANH_WS_FILTER_REGS=0" "-" 2>&1 | sed "s#$P/annonet_amd/##" | tee gpurun_out/call37_ab.txt
