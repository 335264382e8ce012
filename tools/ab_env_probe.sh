#!/bin/bash
# A/B over an environment setting with the short probe (run on the GPU box from the repo root): tools/ab_env_probe.sh "3 4 5" RT_SLICE_LEAVE 1 2 3
# Prints, per value and configuration, ms per frame with one Render per launch and with 16 per launch (tools/cfg5_probe.py).
CFGS=$1; VAR=$2; shift 2
for v in "$@"; do
  for c in $CFGS; do
    env $VAR=$v timeout -k 10 300 python -u tools/cfg5_probe.py $c > gpurun_out/abq_${VAR}_${v}_$c.log 2>&1 || { echo "$VAR=$v cfg$c failed"; tail -3 gpurun_out/abq_${VAR}_${v}_$c.log; exit 1; }
    echo "$VAR=$v cfg$c: $(grep 'per launch' gpurun_out/abq_${VAR}_${v}_$c.log | sed -n '2p;4p' | sed 's/, pre-pass.*//' | tr '\n' '|')"
  done
done
