set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
AB_NO_INFER=1 bash tools/ab_env.sh 4 "-" "ANH_WS_PSTAT=1" "ANH_WS_XCD_BANDS=0" "ANH_WS_PSTAT=8" "ANH_WS_PSTAT=0" 2>&1 | tee gpurun_out/call39_ab.txt
