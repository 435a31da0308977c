#!/usr/bin/env python3
"""A/B of option sets on one library: renders a world twice (MIRT_OPTS_A / MIRT_OPTS_B: name=value,...), checks that the radiance
sums and ray counts are bit-identical, then times both in three interleaved rounds.
python tools/ab_opts.py <world> <w> <h> <spp_check> <spp_time>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracinginaweekend_amd import capi, scene
world, w, h, spp_c, spp_t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mk = {"cornell": lambda: scene.cornell_box(w / h), "cornell2": lambda: scene.cornell_two_boxes(w / h),
      "spheres4": lambda: scene.spheres_world(w / h, small_spheres=False), "spheres": lambda: scene.spheres_world(w / h),
      "spheres1m": lambda: scene.spheres_world(w / h, grid=500, look_from=(26.0, 6.0, 6.0)),
      "spheres90k": lambda: scene.spheres_world(w / h, grid=150, look_from=(26.0, 6.0, 6.0))}[world]
s, cam = mk()
ctxs = []
for tag in ("A", "B"):
    ctx = capi.Context(0)
    ctx.set_option("time_launches", 1)
    for kv in os.environ.get("MIRT_OPTS_" + tag, "").split(","):
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_scene(s); ctx.set_camera(cam)
    ctxs.append(ctx)
res = [c.render(w, h, spp_c, 50, 0x5EED, capi.FLAG_ACCEL_AUTO, want=("sum",)) for c in ctxs]
same = np.array_equal(res[0]["sum"].view(np.uint32), res[1]["sum"].view(np.uint32))
print(f"{world} {w}x{h}@{spp_c}: sums identical: {same}; rays {res[0]['stats'].rays} {res[1]['stats'].rays}", flush=True)
if not same or res[0]["stats"].rays != res[1]["stats"].rays:
    sys.exit(1)
for rnd in range(3):
    for tag, c in zip("AB", ctxs):
        st = c.render(w, h, spp_t, 50, 0x5EED, capi.FLAG_ACCEL_AUTO, want=())["stats"]
        print(f"{tag} [{os.environ.get('MIRT_OPTS_' + tag, '')}] {world} {w}x{h}@{spp_t}: device {st.device_ms:.2f} ms  {st.samples / st.device_ms / 1e3:.1f} Msamples/s  launches {st.extend_ms:.2f} ms", flush=True)
