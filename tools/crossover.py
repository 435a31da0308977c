#!/usr/bin/env python3
"""List walk vs near-first BVH walk by primitive count (the ACCEL_AUTO crossover): random-sphere worlds of growing size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracinginaweekend_amd import capi, scene
w, h, spp = 1200, 800, int(os.environ.get("MIRT_SPP", "16"))
ctx = capi.Context(0)
for grid in (1, 2, 3, 4, 5, 6, 8):
    s, cam = scene.spheres_world(w / h, grid=grid)
    ctx.set_option("lds_prims", 65536)
    ctx.upload_scene(s); ctx.set_camera(cam)
    out = []
    for name, flags in (("list", capi.FLAG_ACCEL_LIST), ("bvh", capi.FLAG_ACCEL_ORDERED)):
        best = 1e9
        for _ in range(3):
            r = ctx.render(w, h, spp, 50, 1, flags, want=())
            best = min(best, r["stats"].device_ms)
        out.append(f"{name} {best:7.2f} ms")
    print(f"grid {grid}: {s.n_prims:4d} primitives   " + "   ".join(out), flush=True)
