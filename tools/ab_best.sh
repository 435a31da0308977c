#!/bin/bash
# Interleaved A/B timing of library variants, best device time of 5 repetitions per round, 4 rounds.
# MIRT_SIZE=WxH bash tools/ab_best.sh <spp> <world> <variant> [<variant> ...]   (libmirt_<variant>.so under raytracinginaweekend_amd/lib/)
L=raytracinginaweekend_amd/lib
spp=$1; world=$2; shift 2
for i in 1 2 3 4; do for v in "$@"; do
  t=$(MIRT_LIB=$L/libmirt_$v.so timeout -k 10 120 python tools/quickbench.py $spp $world 5 | awk '{for(i=1;i<=NF;i++) if($i=="device") print $(i+1)}' | sort -n | head -1) || exit 1
  echo "$world round $i  $v  $t ms"
done; done
