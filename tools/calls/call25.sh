set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/micro/ws_interference.hip -o /tmp/wsi
timeout -k 10 120 /tmp/wsi 2>&1 | tee gpurun_out/call25_wsi.txt
