#!/usr/bin/env python3
"""Wall time of whole mirt_render calls (host work, launches, copies back) beside the device time they report: what an interactive
caller (the viewer loop, examples/flythrough.cpp) pays per preview frame.  python tools/wall_vs_device.py [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracinginaweekend_amd import capi, scene
world = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w, h = (1024, 1024) if world == "cornell" else (1200, 800)
s, cam = scene.cornell_box(1.0) if world == "cornell" else scene.spheres_world(w / h)
ctx = capi.Context(0)
ctx.upload_scene(s); ctx.set_camera(cam)
for spp, flags, want in ((1, capi.FLAG_ACCEL_AUTO | capi.FLAG_ONE_SPP, ("rgba8",)), (1, capi.FLAG_ACCEL_AUTO, ("rgba8",)), (4, capi.FLAG_ACCEL_AUTO, ("rgba8",)),
                         (4, capi.FLAG_ACCEL_AUTO, ("sum", "pixels", "rgba8")), (16, capi.FLAG_ACCEL_AUTO, ("rgba8",))):
    ctx.render(w, h, spp, 50, 1, flags, want=want)
    t = []; d = []
    for rep in range(20):
        t0 = time.perf_counter()
        r = ctx.render(w, h, spp, 50, 1 + rep, flags, want=want)
        t.append(time.perf_counter() - t0); d.append(r["stats"].device_ms)
    t.sort(); d.sort()
    print(f"{world} {w}x{h}@{spp} flags {flags} outputs {'+'.join(want)}: wall {1e3 * t[len(t) // 2]:.3f} ms (min {1e3 * t[0]:.3f}), device {d[len(d) // 2]:.3f} ms", flush=True)
