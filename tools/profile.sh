#!/bin/bash
# rocprofv3 passes over the benchmark (run on the GPU box from the repo root): one kernel-trace/stats run, then one run
# per PMC counter group (counters never share a run with a trace domain; every pass has its own time limit).
set -u
CFG=${1:-2}
KIND=${2:-single}            # single: one Render per launch; throughput: launches of 16 Renders
OUT=gpurun_out/prof_cfg${CFG}_${KIND}
mkdir -p $OUT
export TMPDIR=/tmp
WARM=3; [ "$KIND" = "throughput" ] && WARM=0      # (warm-up frames are single launches: they would mix into the per-kernel means)
BENCH="python3 bench.py --config ${CFG} --steps 16 --warmup ${WARM} --no-cpu-baseline --profile ${KIND}"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($grp) failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/pmc_summary.py $OUT > ${OUT}_summary.json
grep "^{" $OUT/trace.log | tail -1 > ${OUT}_bench.json
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.csv" -size +1M -delete
echo "done $OUT"
