#!/usr/bin/env python3
"""A/B helper: renders a world with the library named by MIRT_LIB and saves the radiance sum + ray count (npz) so that two
builds can be compared bit for bit.  python tools/ab_image.py out.npz [world w h spp]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracinginaweekend_amd import capi, scene
out = sys.argv[1]
world = sys.argv[2] if len(sys.argv) > 2 else "cornell"
w, h, spp = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (256, 256, 16)
s, cam = {"cornell": lambda: scene.cornell_box(w / h), "two_boxes": lambda: scene.cornell_two_boxes(w / h),
          "spheres4": lambda: scene.spheres_world(w / h, small_spheres=False), "spheres": lambda: scene.spheres_world(w / h)}[world]()
with capi.Context(0) as ctx:
    ctx.upload_scene(s); ctx.set_camera(cam)
    r = ctx.render(w, h, spp, 50, 0x5EED, capi.FLAG_ACCEL_LIST if world != "spheres" else capi.FLAG_ACCEL_AUTO, want=("sum",))
    np.savez(out, sum=r["sum"], rays=np.uint64(r["stats"].rays), hist=np.array(list(r["stats"].depth_hist), np.uint64))
    print(world, w, h, spp, "rays", r["stats"].rays, "device ms", round(r["stats"].device_ms, 3))
