#!/bin/bash
# Builds another copy of the library with extra compiler flags, for A/B runs: tools/build_variant.sh NAME "-DRT_COOP_LANES=4"
# -> variants/NAME.so (load it with RT_MI355X_LIB=variants/NAME.so).
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../../variants/obj_$1
F="-O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $2"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c rt_kernels.hip -o ../../variants/obj_$1/k.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c rt_static.hip -o ../../variants/obj_$1/s.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c rt_group.hip -o ../../variants/obj_$1/g.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -x hip -c rt_api.cpp -o ../../variants/obj_$1/a.o
/opt/rocm/bin/hipcc $F -x c++ -c rt_host.cpp -o ../../variants/obj_$1/h.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../variants/$1.so ../../variants/obj_$1/k.o ../../variants/obj_$1/s.o ../../variants/obj_$1/g.o ../../variants/obj_$1/a.o ../../variants/obj_$1/h.o
rm -rf ../../variants/obj_$1
