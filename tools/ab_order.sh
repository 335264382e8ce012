#!/bin/bash
# ms per Render of configurations in a given walk order (run on the GPU box from the repo root): tools/ab_order.sh reference "3 6" [lib]
ORDER=$1; CFGS=$2; LIB=${3:-}
for c in $CFGS; do
  steps=32; [ "$c" = "4" ] && steps=16; [ "$c" = "5" ] && steps=16
  RT_MI355X_LIB=$LIB timeout -k 10 400 python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-other-configs --order $ORDER 2>/dev/null > gpurun_out/abo_$c.json || { echo "cfg$c failed"; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/abo_$c.json').read().strip().splitlines()[-1]); r=d['roofline']
print('cfg$c order=$ORDER lib=$LIB: %.4f ms per Render, kernel %.4f, x16 %.4f ms per frame, view change %.4f' % (d['ms_per_step'],r['kernel_avg_ms'],d['throughput_mode']['ms_per_frame'], d['view_change']['rotate']['ms_per_render']))"
done
