#!/bin/bash
# The C++ host's bench, linked statically with the library and built with AddressSanitizer on the HOST side only (the device code is
# the shipped code), run over the entry points a drop-in uses (run on the GPU box from the repo root after building variants/asan/
# as tools/asan_bench.sh build does here):  tools/asan_bench.sh build | run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  mkdir -p variants/asan; O=$PWD/variants/asan; cd raytracer_amd/csrc
  F="-O1 -g -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
  for f in rt_kernels.hip rt_static.hip rt_group.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c $f -o $O/${f%.hip}.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -x hip -c rt_api.cpp -o $O/rt_api.o
  /opt/rocm/bin/hipcc $F -x c++ -c rt_host.cpp -o $O/rt_host.o
  /opt/rocm/bin/hipcc $F -x c++ -c rt_bench.cpp -o $O/rt_bench.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o $O/rt_bench_asan $O/*.o -pthread
  rm -f $O/*.o
  exit 0
fi
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export RT_BENCH_FAST_EXIT=1
B=variants/asan/rt_bench_asan
timeout -k 10 120 $B --config 2 --frames 8 --warmup 2 && \
timeout -k 10 120 $B --config 2 --frames 32 --frames-per-launch 16 --d2h 2 && \
timeout -k 10 120 $B --config 3 --frames 8 --warmup 2 --d2h 1 && \
timeout -k 10 120 $B --config 3 --frames 32 --frames-per-launch 16 --reuse-first-hits 1 && \
timeout -k 10 120 $B --config 3 --frames 8 --moving 1 && \
timeout -k 10 120 $B --config 3 --frames 8 --devices 0,0,0 --split tiles && \
timeout -k 10 120 $B --config 3 --frames 24 --devices 0,0 --split samples && \
timeout -k 10 120 $B --mode 2 --grid 120 --width 640 --height 360 --bounces 8 --frames 6 && echo "asan runs done"
