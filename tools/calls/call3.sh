set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_schedules.py -x -q > gpurun_out/call3_tests.txt 2>&1 || true
tail -5 gpurun_out/call3_tests.txt
AB_NO_INFER=1 bash tools/ab_env.sh 3 "ANH_WS_PSTAT=1" "ANH_WS_PSTAT=3" "ANH_WS_PSTAT=4" 2>&1 | tee gpurun_out/call3_ab.txt
for v in 1 3 4; do
  ANH_WS_PSTAT=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" > gpurun_out/call3_verbose_$v.txt
done
python - <<'PY'
import re
def load(v):
    d={}
    for l in open(f'gpurun_out/call3_verbose_{v}.txt'):
        p=l.split(); d[p[0]]=float(p[1])
    return d
a,b,c=load(1),load(3),load(4)
for k in a:
    if 'dgrad' in k or 'apply' in k: print(f"{k:48s} {a[k]*1000:7.1f} {b.get(k,0)*1000:7.1f} {c.get(k,0)*1000:7.1f}")
PY
