set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
AB_NO_INFER=1 bash tools/ab_env.sh 8 "ANH_WGRAD_XCD_BANDS=0" "ANH_WGRAD_XCD_BANDS=1" 2>&1 | tee gpurun_out/call53_ab.txt
python - <<'PY'
import re
a,b=[],[]
for l in open('gpurun_out/call53_ab.txt'):
    m=re.search(r'BANDS=(\d)\].*: ([0-9.]+)',l)
    if m: (a if m.group(1)=='0' else b).append(float(m.group(2)))
print(sum(a)/len(a), sum(b)/len(b), sorted(a)[len(a)//2], sorted(b)[len(b)//2])
PY
