set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
for v in 0 1; do
  ANH_LIBRARY=$P/annonet_amd/lib_prof/libannonet_hip.so ANH_WS_TEAMS=$v ANH_WS_PROF=1 timeout -k 10 300 python tools/ws_phase_profile.py 2>&1 | grep -A40 "step 2" | grep "ws prof" | grep "NT=1 kind=[0-9] c_red=32" | cut -c1-330 > gpurun_out/call26_train_$v.txt
  ANH_LIBRARY=$P/annonet_amd/lib_prof/libannonet_hip.so ANH_WS_TEAMS=$v ANH_WS_PROF=1 timeout -k 10 300 python tools/ws_phase_profile.py infer 2>&1 | grep -A400 "infer pass 1" | grep "ws prof" | head -12 | cut -c1-330 > gpurun_out/call26_infer_$v.txt
done
for f in train_0 train_1 infer_0 infer_1; do echo "## $f"; cat gpurun_out/call26_$f.txt; done
