#!/bin/bash
# Same-box comparison of several environment settings: tools/ab_envs.sh ROUNDS "A=1 B=2" "A=0" ...  (each argument one setting; "-" = none;
# ANH_LIBRARY=lib_name selects annonet_amd/lib_name/libannonet_hip.so).  Second column: per-kernel lines matching $AB_SHOW (optional).
rounds="$1"; shift
for r in $(seq 1 "$rounds"); do
  for setting in "$@"; do
    [ "$setting" = "-" ] && envs="" || envs="$setting"
    envs=$(echo "$envs" | sed "s#ANH_LIBRARY=\([A-Za-z0-9_]*\)#ANH_LIBRARY=$PWD/annonet_amd/\1/libannonet_hip.so#")
    if [ "$AB_MODE" = "infer" ]; then   # AB_MODE=infer: the 4096^2 tiled-inference bench instead of the training step
      out=$(env $envs ANH_BENCH_VERBOSE=1 python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>&1)
      echo "[$setting]  $(echo "$out" | grep -o '"value": [0-9.]*' | head -1) Mpx/s  $([ -n "$AB_SHOW" ] && echo "$out" | grep -E "$AB_SHOW" | grep 'ms/step' | awk '{printf "%s %s  ", $1, $2}')"
      continue
    fi
    out=$(env $envs ANH_BENCH_VERBOSE=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1)
    echo "[$setting]  $(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1)  $([ -n "$AB_SHOW" ] && echo "$out" | grep -E "$AB_SHOW" | grep 'ms/step' | awk '{printf "%s %s  ", $1, $2}')"
  done
done
