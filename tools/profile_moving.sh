#!/bin/bash
# rocprofv3 kernel trace of camera-move frames (every Render a first frame after a new rt_set_camera): run on the GPU box from the
# repo root.   tools/profile_moving.sh <config> <rotate|translate>
set -u
CFG=${1:-3}
KIND=${2:-rotate}
OUT=gpurun_out/prof_cfg${CFG}_moving_${KIND}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/view_change_probe.py ${CFG} --steps 16 --only ${KIND} > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
grep "^{" $OUT/trace.log | tail -1 > ${OUT}_bench.json
cp $(find $OUT/trace -name "*_kernel_stats.csv" | head -1) ${OUT}_kernel_stats.csv
find $OUT -name "*_kernel_trace.csv" -delete
echo "done $OUT"
