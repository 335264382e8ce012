#!/bin/bash
# Where the waves' cycles go (issue, wait, by instruction class): tools/pmc_issue.sh CONFIG FRAMES_PER_LAUNCH LAUNCHES TAG
# (run on the GPU box from the repo root; one counter group per rocprofv3 run).
set -u
CFG=$1; PER=$2; N=$3; TAG=${4:-issue}
OUT=gpurun_out/pmc_${TAG}_cfg${CFG}_x${PER}
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 tools/tp_run.py $CFG $PER $N"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- $RUN > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($grp) failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/pmc_summary.py $OUT > ${OUT}_summary.json
find $OUT -name "*.csv" -size +1M -delete
echo done $OUT
