#!/bin/bash
# A/B of an environment switch on ONE box (box-to-box variance is several percent): alternates the settings, three rounds.
# usage: tools/ab_bench.sh VAR valueA valueB [pattern of per-kernel lines to show]
var="$1"; a="$2"; b="$3"; pat="${4:-ms_per_step}"
for r in 1 2 3; do
  for v in "$a" "$b"; do
    out=$(env "$var=$v" ANH_BENCH_VERBOSE=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-infer --prewarm-s 1 2>&1)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$var=$v  $ms"
    echo "$out" | grep -E "$pat" | grep "ms/step" | cut -c1-100
  done
done
