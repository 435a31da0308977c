#!/bin/bash
# A/B of library variants against libmirt_base.so on one box: image identity first (256^2 @16), then three interleaved timing rounds.
# bash tools/ab_gate.sh <spp> <world> <variant> [<variant> ...]    (libmirt_<variant>.so under raytracinginaweekend_amd/lib/)
L=raytracinginaweekend_amd/lib
spp=$1; world=$2; shift 2
iw=$world; [ "$iw" = cornell2 ] && iw=two_boxes
mkdir -p gpurun_out/ab
MIRT_LIB=$L/libmirt_base.so timeout -k 10 100 python tools/ab_image.py gpurun_out/ab/base.npz $iw 256 256 16 || exit 1
for v in "$@"; do
  MIRT_LIB=$L/libmirt_$v.so timeout -k 10 100 python tools/ab_image.py gpurun_out/ab/$v.npz $iw 256 256 16 || exit 1
  echo -n "$v: "; python tools/ab_compare.py gpurun_out/ab/base.npz gpurun_out/ab/$v.npz || exit 1
done
for i in 1 2 3; do for v in base "$@"; do echo -n "$v  "; MIRT_LIB=$L/libmirt_$v.so timeout -k 10 100 python tools/quickbench.py $spp $world 3 | tail -1 || exit 1; done; done
