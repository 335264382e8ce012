#!/bin/bash
# Kernel-trace pass only: per-kernel durations of bench.py for one config and walk order.
#   tools/profile_kernels.sh <config> <order> <tag>
set -u
CFG=${1:-3}; ORDER=${2:-auto}; TAG=${3:-x}
OUT=gpurun_out/kt_cfg${CFG}_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config ${CFG} --order ${ORDER} --steps ${STEPS:-16} --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
cat $(find $OUT -name "*kernel_stats.csv" | head -1)
grep '^{' $OUT/trace.log
