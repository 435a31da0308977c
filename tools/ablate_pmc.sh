#!/bin/bash
# Dynamic instruction attribution: VALU instruction counts of the depth-1 k_bounce launch (Cornell 1024^2, 64 spp,
# max_depth 50; the first k_bounce<false,..> dispatch) for each ablated build of tools/ablate.sh.  Run on the GPU box: bash tools/ablate_pmc.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=raytracinginaweekend_amd/lib
OUT=gpurun_out/prof/ablate
mkdir -p $OUT
for e in 0 1 2 4 8 16 32 14 63; do
  MIRT_LIB=$L/libmirt_exp$e.so MIRT_OPTS=engine=0 MIRT_DEPTH=50 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/e$e -o c -- python3 tools/quickbench.py 64 cornell 1 > $OUT/e$e.log 2>&1 || { echo "variant $e failed"; tail -3 $OUT/e$e.log; exit 1; }
  printf "EXP=%-3s " $e
  python3 - $OUT/e$e/c_results.db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
first = c.execute(f"select min(dispatch_id) from counters_collection where {namecol} like '%k_bounce<false%'").fetchone()[0]
rows = dict(c.execute(f"select counter_name, sum(value) from counters_collection where dispatch_id = {first} group by 1").fetchall())
v, a, t = rows["SQ_INSTS_VALU"], rows["SQ_ACTIVE_INST_VALU"], rows["SQ_THREAD_CYCLES_VALU"]
print(f"VALU {v/1e6:9.1f} M  SALU {rows['SQ_INSTS_SALU']/1e6:8.1f} M  LDS {rows['SQ_INSTS_LDS']/1e6:7.1f} M  lane-util {t/(a*64)*100:5.1f} %  waves {rows['SQ_WAVES']:.0f}")
PY
done
