#!/bin/bash
# Parity of library variants on tree scenes (run on the GPU box from the repo root): tools/ab_parity_tree.sh NAME ...
for v in "$@"; do
  RT_MI355X_LIB=$PWD/variants/$v.so timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread \
    -k "mesh or spheres_and_mesh or sixteen_frames or frames_in_one_launch or zero_direction or first_hit or empty_units or config3" > gpurun_out/abt_$v.log 2>&1 \
    && echo "$v parity ok: $(tail -1 gpurun_out/abt_$v.log)" || { echo "$v parity FAILED"; tail -15 gpurun_out/abt_$v.log; exit 1; }
done
