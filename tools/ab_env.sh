#!/bin/bash
# A/B over an environment setting (run on the GPU box from the repo root): tools/ab_env.sh "3 4" RT_GPOOL 0 1
CFGS=$1; VAR=$2; shift 2
for v in "$@"; do
  for c in $CFGS; do
    steps=20; [ "$c" = "4" ] && steps=10; [ "$c" = "5" ] && steps=10
    env $VAR=$v timeout -k 10 200 python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline > gpurun_out/abe_${VAR}_${v}_$c.log 2>&1 || { echo "$VAR=$v cfg$c failed"; tail -3 gpurun_out/abe_${VAR}_${v}_$c.log; exit 1; }
    echo "$VAR=$v cfg$c $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abe_${VAR}_${v}_$c.log) $(grep -o 'kernel_avg_ms": [0-9.]*' gpurun_out/abe_${VAR}_${v}_$c.log)"
  done
done
