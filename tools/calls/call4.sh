set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
P=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_multidev.py tests/test_gpu_train_main.py tests/test_gpu_infer_main.py tests/test_cpp_shim.py -x -q > gpurun_out/call4_tests.txt 2>&1 || true
tail -5 gpurun_out/call4_tests.txt
for v in 1 3; do
  ANH_WS_PSTAT=$v ANH_WS_PROF=1 ANH_LIBRARY=$P/annonet_amd/lib_prof/libannonet_hip.so python tools/ws_phase_profile.py 2> gpurun_out/call4_phases_$v.txt > /dev/null
  echo "PSTAT=$v"; awk '/--- step 2/{p=1} p && /kind=0/' gpurun_out/call4_phases_$v.txt | cut -c1-260
done
ANH_FUSE_BN_BWD_REDUCE=0 ANH_BENCH_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1 | grep "ms/step\|ms_per_step" | cut -c1-120 > gpurun_out/call4_nofuse.txt
grep "dgrad\|bn_" gpurun_out/call4_nofuse.txt
