set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/micro/conv_bytes.hip -o /tmp/conv_bytes
timeout -k 10 120 /tmp/conv_bytes 2>&1 | tee gpurun_out/call22_conv_bytes.txt
