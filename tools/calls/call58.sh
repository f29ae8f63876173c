set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
true
true
rm -rf gpurun_out/prof
bash tools/collect_profiles.sh > gpurun_out/call58_collect.log 2>&1 || { tail -20 gpurun_out/call58_collect.log; exit 1; }
find gpurun_out/prof -name "*_kernel_trace.csv" -path "*fetch*" -delete; find gpurun_out/prof -name "*_kernel_trace.csv" -path "*write*" -delete
find gpurun_out/prof -name "*_kernel_trace.csv" -path "*/sq/*" -delete; find gpurun_out/prof -name "*_kernel_trace.csv" -path "*ikt*" -delete
find gpurun_out/prof -name "*agent_info*" -delete
du -sh gpurun_out/prof; tail -2 gpurun_out/prof/traffic.txt; tail -1 gpurun_out/prof/infer_traffic.txt; python -c "
import json
d=json.load(open('gpurun_out/prof/bench.json')); print(d['value'], d['ms_per_step'], d.get('value_host_inputs'), d['critical_path']); print(d['cpu_baseline']); i=d['infer']; print(i['value'], i['value_labels_on_host'], i['cpu_baseline'])"
