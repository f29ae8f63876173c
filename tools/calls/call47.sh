set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for r in 1 2 3 4 5; do for v in 0 1 2; do
  i=$(ANH_WS_TALL=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "[ANH_WS_TALL=$v] $i"
done; done 2>&1 | tee gpurun_out/call47_ab.txt
