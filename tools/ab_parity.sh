#!/bin/bash
# Parity of library variants (run on the GPU box from the repo root): tools/ab_parity.sh NAME ... runs the sphere parity tests
# against variants/NAME.so.
for v in "$@"; do
  RT_MI355X_LIB=$PWD/variants/$v.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread \
    -k "default_sphere or bounce_counts or refraction or config2 or ragged or many_spheres or spheres_and_mesh or ray_generation" > gpurun_out/abp_$v.log 2>&1 \
    && echo "$v parity ok: $(tail -1 gpurun_out/abp_$v.log)" || { echo "$v parity FAILED"; tail -15 gpurun_out/abp_$v.log; }
done
