set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_schedules.py -x -q > gpurun_out/call48_tests.txt 2>&1 || true
tail -4 gpurun_out/call48_tests.txt
grep -q "failed\|rror" gpurun_out/call48_tests.txt && exit 1
ANH_ACT_MATERIALIZE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_first_step_dirty_memory.py tests/test_gpu_trainer_state.py -x -q > gpurun_out/call48_tests2.txt 2>&1 || true
tail -4 gpurun_out/call48_tests2.txt
grep -q "failed\|rror" gpurun_out/call48_tests2.txt && exit 1
AB_NO_INFER=1 bash tools/ab_env.sh 5 "ANH_ACT_MATERIALIZE=0" "ANH_ACT_MATERIALIZE=1" 2>&1 | tee gpurun_out/call48_ab.txt
for v in 0 1; do ANH_ACT_MATERIALIZE=$v ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step" > gpurun_out/call48_v$v.txt; done
python - <<'PY'
def load(f):
    d={}
    for l in open(f):
        p=l.split(); d[p[0]]=float(p[1])
    return d
a,b=load('gpurun_out/call48_v0.txt'),load('gpurun_out/call48_v1.txt')
for k in a:
    if abs(a[k]-b.get(k,0))*1000>=2: print(f"{k:52s} {a[k]*1000:8.1f} {b.get(k,0)*1000:8.1f}")
PY
