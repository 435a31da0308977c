#!/bin/bash
# L2 (TCC) hit/miss of one render: bash tools/pmc_tcc.sh [spp] [tag]   (WORLD=spheres1m ...)
SPP=${1:-16}; TAG=${2:-tcc}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/tcc -o c -- python3 tools/quickbench.py $SPP ${WORLD:-cornell} 1 > $OUT/tcc.log 2>&1 || { echo "pass failed"; tail -3 $OUT/tcc.log; exit 1; }
echo "done: $OUT"
