#!/bin/bash
# Cost of the near-tie distance (run on the GPU box from the repo root): configs 3 and 4 with the distance scaled down to about
# what a flat 2^-19 t would be, and at its guaranteed value.
for scale in 0.004 1; do
  for c in 3 4; do
    RT_TIE_SCALE=$scale timeout -k 10 300 python tools/cfg5_probe.py $c > gpurun_out/tie_${scale}_$c.log 2>&1 || { echo "scale $scale cfg $c failed"; tail -3 gpurun_out/tie_${scale}_$c.log; exit 1; }
    echo "scale $scale cfg $c"; grep "per launch" gpurun_out/tie_${scale}_$c.log
  done
done
