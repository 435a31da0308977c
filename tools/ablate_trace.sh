#!/bin/bash
# Builds ablated variants of libmirt for k_trace (MIRT_TEXP bits of rt_device.h, timing only: results are wrong by
# design) here on the CPU; run on the GPU box with `bash tools/ablate_trace.sh run`.
cd "$(dirname "$0")/.."
L=raytracinginaweekend_amd/lib
V="${MIRT_VARIANTS:-0 1 2 4 8 3 7}"
if [ "$1" = "run" ]; then
  for e in $V; do
    printf "TEXP=%-3s " $e
    MIRT_LIB=$L/libmirt_texp$e.so timeout -k 10 100 python tools/quickbench.py 256 cornell 2 | tail -1
  done
else
  for e in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -ldl -pthread -DMIRT_TEXP=$e raytracinginaweekend_amd/csrc/mirt.hip raytracinginaweekend_amd/csrc/mirt_multi.hip -o $L/libmirt_texp$e.so &
  done
  wait
fi
