#!/bin/bash
# Per-kernel times of tools/quickbench.py under two option sets: bash tools/prof_opts.sh <lib under lib/> <spp> <world> <optsA> <optsB>
export TMPDIR=/tmp
L=raytracinginaweekend_amd/lib
for o in "$4" "$5"; do
  tag=$(echo "$o" | tr '=,' '__'); out=gpurun_out/prof/po_$tag; mkdir -p $out
  MIRT_LIB=$L/$1 MIRT_OPTS=$o timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $out -o t -- python3 tools/quickbench.py $2 $3 2 > $out/run.log 2>&1 || { echo "profile failed"; tail -3 $out/run.log; exit 1; }
  echo "== $o"; tail -1 $out/run.log
  python3 tools/rocpd_summary.py $(ls $out/*.db | head -1) | head -7
done
