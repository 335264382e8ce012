#!/bin/bash
# A/B over an environment setting with the benchmark (run on the GPU box from the repo root): tools/ab_env_bench.sh "3 4" RT_SLICE_LEAVE 1 2 3
CFGS=$1; VAR=$2; shift 2
for c in $CFGS; do
  steps=32; [ "$c" = "4" ] && steps=16; [ "$c" = "5" ] && steps=16
  for v in "$@"; do
    env $VAR=$v timeout -k 10 400 python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/abe_${VAR}_${v}_$c.log 2>/dev/null || { echo "$VAR=$v cfg$c failed"; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/abe_${VAR}_${v}_$c.log').read().strip().splitlines()[-1])
print('cfg$c $VAR=$v: %.4f ms per Render (kernel %.4f), x16 %.4f ms per frame' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['throughput_mode']['ms_per_frame']))"
  done
done
