#!/bin/bash
# Same-box comparison of environment settings on the training bench and the inference bench: alternates them for the given number of
# rounds.   usage: tools/ab_env.sh rounds "VAR=a" "VAR=b OTHER=c" ...      ("-" = no setting; ANH_LIBRARY=... selects another build)
rounds="$1"; shift
for r in $(seq $rounds); do
  for s in "$@"; do
    set_=""; [ "$s" != "-" ] && set_="$s"
    t=$(env $set_ python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-infer --prewarm-s 1 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    i=""
    [ -z "$AB_NO_INFER" ] && i=$(env $set_ python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline --prewarm-s 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "[$s]  train $t  infer $i Mpx/s"
  done
done
