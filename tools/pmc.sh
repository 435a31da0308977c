#!/bin/bash
# Collects rocprofv3 counters for one Cornell 1024^2 render of $1 spp (default 32) in separate passes.
# Usage (on the GPU box, from the repo root): bash tools/pmc.sh [spp] [tag]
SPP=${1:-32}; TAG=${2:-pmc}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o c -- python3 tools/quickbench.py $SPP ${WORLD:-cornell} 1 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM &&
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
timeout -s KILL 120 rocprofv3 --kernel-trace -d $OUT/kt -o c -- python3 tools/quickbench.py $SPP ${WORLD:-cornell} 1 > $OUT/kt.log 2>&1
echo "pmc done: $OUT"
