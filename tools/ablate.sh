#!/bin/bash
# Builds ablated variants of libmirt (MIRT_EXP bits, timing only: results are wrong by design) here on the
# CPU; run on the GPU box with `bash tools/ablate.sh run`.  Reports the depth-1 launch time per variant.
cd "$(dirname "$0")/.."
L=raytracinginaweekend_amd/lib
if [ "$1" = "run" ]; then
  for e in 0 1 2 4 8 16 32 6 14 63; do
    printf "EXP=%-3s " $e
    MIRT_LIB=$L/libmirt_exp$e.so MIRT_PER_DEPTH=1 timeout -k 10 100 python tools/quickbench.py 128 cornell 2 | tail -1 | sed 's/  2:.*//'
  done
else
  for e in 0 1 2 4 8 16 32 6 14 63; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -DMIRT_EXP=$e raytracinginaweekend_amd/csrc/mirt.hip -o $L/libmirt_exp$e.so &
  done
  wait
fi
