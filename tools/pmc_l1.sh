#!/bin/bash
# Texture-addresser / L1 (TA, TCP) load of one render, separate passes: bash tools/pmc_l1.sh [spp] [tag]   (WORLD=spheres1m ...)
SPP=${1:-16}; TAG=${2:-l1}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
run() { local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o c -- python3 tools/quickbench.py $SPP ${WORLD:-cornell} 1 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
}
run ta TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum &&
run grbm GRBM_GUI_ACTIVE
echo "done: $OUT"
