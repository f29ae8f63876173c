#!/bin/bash
# Same-box comparison of several library builds (annonet_amd/lib_<name>/libannonet_hip.so; "lib" = the default build) on the training bench
# and the inference bench: alternates them for the given number of rounds.   usage: tools/ab_multi.sh rounds name1 name2 ...
rounds="$1"; shift
for r in $(seq $rounds); do
  for n in "$@"; do
    so="$PWD/annonet_amd/$n/libannonet_hip.so"
    t=$(ANH_LIBRARY="$so" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    i=""; [ -z "$AB_NO_INFER" ] && i=$(ANH_LIBRARY="$so" python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "$n  train $t  infer $i Mpx/s"
  done
done
