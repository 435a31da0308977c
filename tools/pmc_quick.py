#!/usr/bin/env python3
"""Sums the counters of the k_trace* launches found in the rocprofv3 --pmc outputs under the given directories and prints
them per 64 rays.  python tools/pmc_quick.py <rays of the launch> <dir> [<dir> ...]"""
import glob, sqlite3, sys
rays = float(sys.argv[1])
for d in sys.argv[2:]:
    c = {}
    for db in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        c.update(dict(con.execute(f"select counter_name, sum(value) from counters_collection where {namecol} like '%k_trace%' group by 1").fetchall()))
    print("==", d)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if k in c: print(f"  {k:24s} per 64 rays {c[k] * 64 / rays:8.1f}")
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(f"  lane use {100 * c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64):.1f} %")
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if k in c: print(f"  {k:24s} / wave-cycles {100 * c[k] / wc:6.1f} %")
        print(f"  waves {c.get('SQ_WAVES', 0):.0f}  wave-cycles {wc:.3e}  busy-cycles {c.get('SQ_BUSY_CYCLES', 0):.3e}")
