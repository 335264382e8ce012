#!/bin/bash
# A/B over library variants (run on the GPU box from the repo root): tools/ab.sh "3 4" base coop4 ...
# Prints, per variant and configuration, ms per frame with one Render per launch and with 16 per launch (tools/cfg5_probe.py).
CFGS=$1; shift
for v in "$@"; do
  for c in $CFGS; do
    RT_MI355X_LIB=$PWD/variants/$v.so timeout -k 10 300 python -u tools/cfg5_probe.py $c > gpurun_out/ab_${v}_$c.log 2>&1 || { echo "$v cfg$c failed"; tail -3 gpurun_out/ab_${v}_$c.log; exit 1; }
    echo "$v cfg$c: $(grep 'per launch' gpurun_out/ab_${v}_$c.log | sed -n '2p;4p' | sed 's/, pre-pass.*//' | tr '\n' '|')"
  done
done
