#!/bin/bash
# A/B over library variants with the benchmark itself (run on the GPU box from the repo root): tools/ab_bench.sh "3 4" base t21 ...
# Prints, per variant and configuration, ms per Render and ms per frame in launches of 16 (two runs each).
CFGS=$1; shift
for c in $CFGS; do
  steps=32; [ "$c" = "4" ] && steps=16; [ "$c" = "5" ] && steps=16
  for rep in 1 2; do
    for v in "$@"; do
      RT_MI355X_LIB=$PWD/variants/$v.so timeout -k 10 400 python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/abb_${v}_$c.log 2>/dev/null || { echo "$v cfg$c failed"; exit 1; }
      python3 -c "
import json,sys
d=json.loads(open('gpurun_out/abb_${v}_$c.log').read().strip().splitlines()[-1])
print('cfg$c %-6s run $rep: %.4f ms per Render (kernel %.4f), x16 %.4f ms per frame' % ('$v', d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['throughput_mode']['ms_per_frame']))"
    done
  done
done
