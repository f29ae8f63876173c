set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for r in 1 2; do for b in 0 3 5 7 9 13; do
  e=""; [ "$b" != "0" ] && e="ANH_INFER_TILE_BATCH=$b"
  i=$(env $e python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "tile batch $b: $i"
done; done 2>&1 | tee gpurun_out/call10_batch.txt
