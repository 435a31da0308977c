#!/bin/bash
# sweep the slice threshold (walk_yield) and the leaf-pass threshold (leaf_wait) of the near-first walks; best of 5 repetitions each
run() { MIRT_SIZE=$1 MIRT_OPTS=walk_yield=$4,leaf_wait=$5 timeout -k 10 90 python3 tools/quickbench.py $2 $3 5 | awk '{for(i=1;i<=NF;i++) if($i=="device") print $(i+1)}' | sort -n | head -1; }
for y in ${SWEEP_Y1:-40 44 48 52}; do for l in ${SWEEP_L1:-5 6 8 10}; do
  echo "spheres yield=$y leaf=$l: $(run 1200x800 128 spheres $y $l) ms" || exit 1
done; done
for y in ${SWEEP_Y2:-50 52 54}; do for l in ${SWEEP_L2:-4 5 6}; do
  echo "spheres1m yield=$y leaf=$l: $(run 1920x1080 32 spheres1m $y $l) ms" || exit 1
done; done
