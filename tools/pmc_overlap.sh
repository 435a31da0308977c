#!/bin/bash
# Do scalar instructions issue beside vector ones?  One SQ pass: bash tools/pmc_overlap.sh <spp> <world>
# Prints the per-class active cycles next to SQ_ACTIVE_INST_ANY and SQ_BUSY_CYCLES: a sum of the classes above ANY is overlap.
export TMPDIR=/tmp
OUT=gpurun_out/prof/overlap_$2; mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT -o c -- python3 tools/quickbench.py ${1:-128} ${2:-cornell} 1 > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import sqlite3, sys
out = sys.argv[1]
c = sqlite3.connect(out + "/c_results.db")
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
d = dict(c.execute(f"select counter_name, sum(value) from counters_collection where {namecol} like '%k_trace%' group by 1").fetchall())
any_ = d["SQ_ACTIVE_INST_ANY"]
parts = ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"]
print(out.split("overlap_")[-1])
for k in ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY"] + parts:
    print(f"  {k:22s} {d[k]:.4e}   {d[k] / any_:6.3f} x ANY")
print(f"  sum of classes / ANY = {sum(d[k] for k in parts) / any_:.3f}")
PY
