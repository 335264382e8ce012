#!/bin/bash
# PMC passes over tools/sorted_rays_probe.py --pmc (scattered rays of one frame, persistent list walk): VALU lane use and L1 / L2
# hit rates of the walk in path order (rt_probe_walk_kernel<0>) and sorted by octant + origin cell (<1>).  Run on the GPU box.
set -u
CFG=${1:-4}
OUT=gpurun_out/prof_sorted_cfg${CFG}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 tools/sorted_rays_probe.py ${CFG} --pmc --repeats 2 > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($grp) failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/pmc_summary.py $OUT > ${OUT}_summary.json
find $OUT -name "*.csv" -size +1M -delete
echo "done $OUT"
