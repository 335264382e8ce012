#!/bin/bash
# PMC passes over launches of one kind (run on the GPU box from the repo root): tools/pmc_clean.sh CONFIG FRAMES_PER_LAUNCH LAUNCHES TAG
# One counter group per rocprofv3 run, each with its own time limit; summary JSON next to the raw output.
set -u
CFG=$1; PER=$2; N=$3; TAG=${4:-clean}
OUT=gpurun_out/pmc_${TAG}_cfg${CFG}_x${PER}
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 tools/tp_run.py $CFG $PER $N"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- $RUN > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($grp) failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/pmc_summary.py $OUT > ${OUT}_summary.json
find $OUT -name "*.csv" -size +1M -delete
echo done $OUT
