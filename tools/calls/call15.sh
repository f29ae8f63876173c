set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python - <<'PY'
import os, subprocess, sys, numpy as np, tempfile
here='tests/helpers/run_train_steps.py'
outs={}
for v in ('1','8','7'):
    o=tempfile.mktemp(suffix='.npz')
    r=subprocess.run([sys.executable, here, o], env=dict(os.environ, ANH_WS_PSTAT=v), capture_output=True, text=True)
    assert r.returncode==0, r.stderr[-500:]
    outs[v]=np.load(o)
for v in ('8','7'):
    print('PSTAT', v, 'vs 1: params identical', np.array_equal(outs[v]['params'], outs['1']['params']), 'losses identical', np.array_equal(outs[v]['losses'], outs['1']['losses']),
          'rel', float(np.linalg.norm(outs[v]['params']-outs['1']['params'])/np.linalg.norm(outs['1']['params'])))
PY
AB_NO_INFER=1 bash tools/ab_env.sh 5 "ANH_WS_PSTAT=1" "ANH_WS_PSTAT=8" 2>&1 | tee gpurun_out/call15_ab.txt
