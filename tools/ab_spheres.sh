#!/bin/bash
# A/B over library variants on the sphere configuration (run on the GPU box from the repo root): tools/ab_spheres.sh base pair ...
for v in "$@"; do
  RT_MI355X_LIB=$PWD/variants/$v.so timeout -k 10 200 python bench.py --config 2 --steps 64 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/abs_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/abs_$v.log; exit 1; }
  python3 - "$v" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/abs_%s.log" % sys.argv[1]) if l.startswith("{")][-1])
print("%s: %.4f ms/step, kernel %.4f ms, frac %.4f | 16 per launch: %.4f ms/frame, frac %.4f" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_avg_ms"],
      d["roofline"]["frac"], d["throughput_mode"]["ms_per_frame"], d["throughput_mode"]["roofline_frac"]))
PY
done
