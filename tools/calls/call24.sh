set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for v in 0 1; do
  ANH_WS_TEAMS=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" > gpurun_out/call24_train_$v.txt
  ANH_WS_TEAMS=$v ANH_CONCURRENT_WGRAD=0 ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" > gpurun_out/call24_train1s_$v.txt
  ANH_WS_TEAMS=$v ANH_BENCH_VERBOSE=1 python bench.py --mode infer --steps 5 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>&1 | grep "ms/image" > gpurun_out/call24_infer_$v.txt
done
python - <<'PY'
def load(f):
    d={}
    for l in open(f):
        p=l.split(); d[p[0]]=float(p[1])
    return d
for tag in ("train","train1s","infer"):
    a,b=load(f'gpurun_out/call24_{tag}_0.txt'),load(f'gpurun_out/call24_{tag}_1.txt')
    print("##",tag)
    for k in a:
        if abs(a[k]-b.get(k,0))*1000>=2 or 'L8' in k: print(f"{k:52s} {a[k]*1000:8.1f} {b.get(k,0)*1000:8.1f}")
PY
