#!/bin/bash
# One SQ counter pass over tools/quickbench.py for the library named by $1 (tag $2): instruction counts, lane use, waits.
export TMPDIR=/tmp
OUT=gpurun_out/prof/ab_$2; mkdir -p $OUT
MIRT_LIB=$1 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d $OUT -o c -- python3 tools/quickbench.py ${3:-256} cornell 1 > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/run.log; exit 1; }
python3 - <<PY
import sqlite3
c = sqlite3.connect("$OUT/c_results.db")
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
d = dict(c.execute(f"select counter_name, sum(value) from counters_collection where {namecol} like '%k_trace%' group by 1").fetchall())
print("$2", {k: f"{v:.4g}" for k, v in sorted(d.items())})
print("   lane use %.1f %%  waiting %.1f %%  issue-stalled %.1f %%" % (100 * d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]), 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]))
PY
