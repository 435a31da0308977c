#!/bin/bash
# One SQ counter pass over tools/quickbench.py for a library: bash tools/pmc_ab2.sh <lib.so under lib/> <tag> <spp> <world> [size]
# Prints VALU / SALU / LDS / VMEM instructions per 64 rays of the k_trace* launch, lane use and the waiting share.
export TMPDIR=/tmp
L=raytracinginaweekend_amd/lib
OUT=gpurun_out/prof/ab2_$2; mkdir -p $OUT
MIRT_SIZE=${5:-} MIRT_LIB=$L/$1 timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT -o c -- python3 tools/quickbench.py ${3:-128} ${4:-spheres} 1 > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import re, sqlite3, sys
out = sys.argv[1]
line = [l for l in open(out + "/run.log") if "Msamples/s" in l][-1]
w, h, spp = (int(x) for x in re.search(r"(\d+)x(\d+)@(\d+)", line).groups())
rays = w * h * spp * float(re.search(r"rays/sample ([\d.]+)", line).group(1))
c = sqlite3.connect(out + "/c_results.db")
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
d = dict(c.execute(f"select counter_name, sum(value) from counters_collection where {namecol} like '%k_trace%' group by 1").fetchall())
per = lambda k: d[k] * 64 / rays
print(f"{out.split('ab2_')[-1]:14s} per 64 rays: VALU {per('SQ_INSTS_VALU'):7.1f}  SALU {per('SQ_INSTS_SALU'):7.1f}  LDS {per('SQ_INSTS_LDS'):6.1f}  VMEM_RD {per('SQ_INSTS_VMEM_RD'):6.1f}   lane use {100 * d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU']):.1f} %  waiting {100 * d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.1f} %  wave-cycles {d['SQ_WAVE_CYCLES']:.3e}")
PY
