set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
for ns in 0 1; do
  ANH_WS_PROF_NOSTORE=$ns ANH_WS_PROF=1 ANH_LIBRARY=$P/annonet_amd/lib_prof/libannonet_hip.so python tools/ws_phase_profile.py 2> gpurun_out/call8_train_$ns.txt > /dev/null
  echo "== training step, nostore=$ns"; awk '/--- step 2/{p=1} p && /ws prof/' gpurun_out/call8_train_$ns.txt | sed 's/\[ws prof\] //; s/ | kernel.*loop / | loop /; s/(+ drain.*//' | cut -c1-230
  ANH_WS_PROF_NOSTORE=$ns ANH_WS_PROF=1 ANH_LIBRARY=$P/annonet_amd/lib_prof/libannonet_hip.so python tools/ws_phase_profile.py infer 2> gpurun_out/call8_infer_$ns.txt > /dev/null
  echo "== inference, nostore=$ns"; awk '/--- infer pass 1/{p=1} p && /ws prof/' gpurun_out/call8_infer_$ns.txt | head -9 | sed 's/\[ws prof\] //; s/ | kernel.*loop / | loop /; s/(+ drain.*//' | cut -c1-230
done
