set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/micro/simd_sharing.hip -o /tmp/simd_sharing
timeout -k 10 120 /tmp/simd_sharing 2>&1 | tee gpurun_out/call29_simd_sharing.txt
