#!/bin/bash
# rt_host.cpp (the KD builder and the camera, no GPU code) under AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU:
#   tools/host_sanitizers.sh [rounds]
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/rt_san
for kind in address,undefined thread; do
  g++ -std=c++17 -O1 -g -fsanitize=$kind -fno-omit-frame-pointer -Iinclude tests/cpp/host_sanitizers.cpp raytracer_amd/csrc/rt_host.cpp -o /tmp/rt_san/san_${kind%%,*} -pthread
  echo "== -fsanitize=$kind"
  /tmp/rt_san/san_${kind%%,*} ${1:-6} | tail -n 4
done
