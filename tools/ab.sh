#!/bin/bash
# A/B over library variants (run on the GPU box from the repo root): tools/ab.sh "3 4" base coop4 ...
CFGS=$1; shift
for v in "$@"; do
  for c in $CFGS; do
    steps=20; [ "$c" = "4" ] && steps=10; [ "$c" = "5" ] && steps=10
    RT_MI355X_LIB=$PWD/variants/$v.so timeout -k 10 200 python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline > gpurun_out/ab_${v}_$c.log 2>&1 || { echo "$v cfg$c failed"; tail -3 gpurun_out/ab_${v}_$c.log; exit 1; }
    echo "$v cfg$c $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_${v}_$c.log) $(grep -o 'kernel_avg_ms": [0-9.]*' gpurun_out/ab_${v}_$c.log)"
  done
done
