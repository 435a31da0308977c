#!/usr/bin/env python3
"""Shader clock of the persistent kernel's waves, from the dump of a -DMIRT_TEXP=80 build (16: per-wave dump, 64: shader-clock
cycles in place of the hardware ids; MIRT_WAVE_DUMP=file): s_memtime cycles / s_memrealtime (100 MHz) per wave.
python tools/wave_clock.py file"""
import sys
import numpy as np
w = np.fromfile(sys.argv[1], np.uint32).reshape(-1, 8)
w = w[w[:, 1] > 0]
cyc = (w[:, 2].astype(np.uint64) | (w[:, 3].astype(np.uint64) << np.uint64(32))).astype(np.float64)
t0 = (w[:, 4].astype(np.uint64) | (w[:, 5].astype(np.uint64) << np.uint64(32))).astype(np.float64)
t1 = (w[:, 6].astype(np.uint64) | (w[:, 7].astype(np.uint64) << np.uint64(32))).astype(np.float64)
us = (t1 - t0) / 100.0
mhz = cyc / us
print(f"{len(w)} waves, kernel {(t1.max() - t0.min()) / 100:.1f} us; shader clock per wave: median {np.median(mhz):.0f} MHz, p5 {np.percentile(mhz, 5):.0f}, p95 {np.percentile(mhz, 95):.0f}; "
      f"iterations per wave: mean {w[:, 1].mean():.0f}; cycles per loop iteration per wave: {np.median(cyc / w[:, 1]):.0f}")
