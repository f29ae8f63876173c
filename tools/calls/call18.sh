set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for r in 1 2 3; do for v in 0 2 3 4; do
  i=$(ANH_WS_INFER_PSTORE=$v python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "INFER_PSTORE=$v infer $i"
done; done 2>&1 | tee gpurun_out/call18_ab.txt
