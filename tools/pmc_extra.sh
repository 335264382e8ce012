#!/bin/bash
# Texture-address / L1 pipeline counters for one benchmark configuration (run on the GPU box from the repo root).
set -u
CFG=${1:-3}
OUT=gpurun_out/pmcx_cfg${CFG}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --config ${CFG} --steps 8 --warmup 2 --no-cpu-baseline"
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/pmc_summary.py $OUT
