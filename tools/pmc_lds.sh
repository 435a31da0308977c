#!/bin/bash
# LDS counters of one k_trace* launch for a library: bash tools/pmc_lds.sh <lib.so under lib/> <tag> <spp> <world> [size]
export TMPDIR=/tmp
L=raytracinginaweekend_amd/lib
OUT=gpurun_out/prof/lds_$2; mkdir -p $OUT
MIRT_SIZE=${5:-} MIRT_LIB=$L/$1 timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT -o c -- python3 tools/quickbench.py ${3:-128} ${4:-cornell} 1 > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import sqlite3, sys
out = sys.argv[1]
c = sqlite3.connect(out + "/c_results.db")
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
d = dict(c.execute(f"select counter_name, sum(value) from counters_collection where {namecol} like '%k_trace%' group by 1").fetchall())
line = [l for l in open(out + "/run.log") if "Msamples/s" in l][-1].strip()
print(out.split("lds_")[-1], "|", line)
print("   " + "  ".join(f"{k} {v:.4g}" for k, v in sorted(d.items())))
print(f"   lds_bank_conflict_rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {100 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.1f} %;  conflict cycles / wave-cycles = {100 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_WAVE_CYCLES']:.2f} %;  waiting on LDS / wave-cycles = {100 * d['SQ_WAIT_INST_LDS'] / d['SQ_WAVE_CYCLES']:.2f} %")
PY
