set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/call5_tests.txt 2>&1 || true
tail -8 gpurun_out/call5_tests.txt
for w in 1 0; do for d in 0,0 0,0,0,0,0,0,0,0; do
  echo "ANH_REPLICA_WORKERS=$w devices=$d"
  ANH_REPLICA_WORKERS=$w python bench.py --in-process --devices $d --steps 40 --warmup 5 --prewarm-s 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('replicas','value','ms_per_step','host_us_per_start_training','worker_wakeups_per_step')}, d['exchange'])"
done; done 2>&1 | tee gpurun_out/call5_inprocess.txt
