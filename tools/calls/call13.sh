set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash tools/ab_env.sh 3 "ANH_WS_STAGGER=0" "ANH_WS_STAGGER=8" "ANH_WS_STAGGER=16" "ANH_WS_STAGGER=32" 2>&1 | tee gpurun_out/call13_ab.txt
