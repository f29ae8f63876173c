set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/call57_tests.txt 2>&1 || true
tail -8 gpurun_out/call57_tests.txt
