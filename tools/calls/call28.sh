set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
AB_NO_INFER=1 bash tools/ab_env.sh 5 "ANH_WS_TEAMS=0" "ANH_WS_TEAMS=3" "ANH_WS_TEAMS=4" 2>&1 | tee gpurun_out/call28_ab.txt
