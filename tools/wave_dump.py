#!/usr/bin/env python3
"""Reads the per-wave schedule dump of a -DMIRT_TEXP=16 build (MIRT_WAVE_DUMP=file): per wave the path ids it generated, its
loop iterations, HW_ID, XCC_ID and start/end times (100 MHz clock).  Prints how evenly the waves of the persistent grid
advanced.  python tools/wave_dump.py file"""
import sys
import numpy as np
w = np.fromfile(sys.argv[1], np.uint32).reshape(-1, 8)
w = w[w[:, 1] > 0]
made, iters, hw, xcc = w[:, 0].astype(np.float64), w[:, 1].astype(np.float64), w[:, 2], w[:, 3] & 0xF
t0 = (w[:, 4].astype(np.uint64) | (w[:, 5].astype(np.uint64) << np.uint64(32))).astype(np.float64)
t1 = (w[:, 6].astype(np.uint64) | (w[:, 7].astype(np.uint64) << np.uint64(32))).astype(np.float64)
base = t0.min()
dur = (t1 - t0) / 100.0                                   # microseconds
print(f"{len(w)} waves; start spread {(t0.max() - base) / 100:.1f} us, end spread {(t1.max() - t1.min()) / 100:.1f} us, kernel {(t1.max() - base) / 100:.1f} us")
rate = made / dur
print(f"paths per wave: mean {made.mean():.0f}  min {made.min():.0f}  max {made.max():.0f};  paths/us per wave: mean {rate.mean():.3f} min {rate.min():.3f} max {rate.max():.3f}  (p5 {np.percentile(rate,5):.3f}, p95 {np.percentile(rate,95):.3f})")
late = t0 - base > 0.1 * (t1.max() - base)
print(f"waves that started after 10 % of the kernel: {late.sum()}")
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc.astype(np.int64) * 1000 + se * 100 + sh * 20 + cu
ids, cnt = np.unique(key, return_counts=True)
print(f"{len(ids)} distinct (xcc,se,sh,cu); waves per CU: " + ", ".join(f"{c}:{(cnt == c).sum()}" for c in np.unique(cnt)))
for x in np.unique(xcc):
    m = xcc == x
    print(f"  xcc {x}: {m.sum()} waves, {made[m].sum() / 1e6:.2f} M paths, rate/wave {rate[m].mean():.3f}")
per_cu = {k: rate[key == k].mean() for k in ids}
byc = {}
for k, c in zip(ids, cnt): byc.setdefault(c, []).append(per_cu[k])
for c in sorted(byc): print(f"  CUs with {c} waves: mean rate per wave {np.mean(byc[c]):.3f}")
