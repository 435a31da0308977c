#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs: per-kernel stats of a --kernel-trace run and
per-kernel counter sums of --pmc runs.  Usage: rocpd_summary.py kt.db [pmc.db ...] > summary.txt"""
import sqlite3
import sys


def short(name):
    return name.replace("void mirt::", "").split("(")[0][:70]


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"## kernel trace: {db}")
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
    for n, k, s, a, mn, mx in rows:
        print(f"{short(n):72s} {k:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")


def pmc(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    print(f"## counters: {db}")
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {namecol}, counter_name, count(*), sum(value) from counters_collection group by 1,2 order by 1,2").fetchall()
    print(f"{'kernel':72s} {'counter':26s} {'dispatches':>10s} {'sum':>22s}")
    for n, cn, k, v in rows:
        print(f"{short(n):72s} {cn:26s} {k:10d} {v:22.0f}")


if __name__ == "__main__":
    kernel_stats(sys.argv[1])
    for d in sys.argv[2:]:
        pmc(d)
