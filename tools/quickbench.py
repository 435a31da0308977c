#!/usr/bin/env python3
"""Quick A/B helper for kernel work: Cornell 1024^2 at a few spp, device time and Msamples/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracinginaweekend_amd import capi, scene

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
world = sys.argv[2] if len(sys.argv) > 2 else "cornell"
w, h = (1024, 1024) if world.startswith("cornell") else (1200, 800)
if os.environ.get("MIRT_SIZE"):
    w, h = (int(x) for x in os.environ["MIRT_SIZE"].split("x"))
if world == "cornell":
    s, cam = scene.cornell_box(1.0)
elif world == "cornell2":                     # the reference's Cornell box with its two rotated boxes (18 primitives)
    s, cam = scene.cornell_two_boxes(1.0)
elif world == "textured":                     # two 1024^2 image textures (4 MB each) on floor, wall and ball
    from tests import helpers as H
    s, cam = scene.textured_world(w / h, H.synthetic_texture(1024, 1024, 1, "ramp"), H.synthetic_texture(1024, 1024, 2, "checker"))
elif world == "spheres4":                     # BASELINE config 0: ground + three big spheres
    s, cam = scene.spheres_world(w / h, small_spheres=False)
elif world == "spheres1m":                    # BASELINE config 4: ~1M spheres, camera raised to see the field
    s, cam = scene.spheres_world(w / h, grid=500, look_from=(26.0, 6.0, 6.0))
else:
    s, cam = scene.spheres_world(w / h)
ctx = capi.Context(0)
ctx.set_option("time_launches", 1)
for kv in os.environ.get("MIRT_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_scene(s); ctx.set_camera(cam)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for rep in range(reps):
    r = ctx.render(w, h, spp, int(os.environ.get('MIRT_DEPTH', '50')), 0x5EED, capi.FLAG_ACCEL_AUTO, want=())
    st = r["stats"]
    print(f"{world} {w}x{h}@{spp}: device {st.device_ms:.2f} ms  {st.samples / st.device_ms / 1e3:.1f} Msamples/s  "
          f"{st.rays / st.device_ms / 1e6:.2f} Grays/s  first {st.first_ms:.2f} ms  deeper {st.extend_ms:.2f} ms  rays/sample {st.rays / st.samples:.3f}", flush=True)
    if os.environ.get("MIRT_PER_DEPTH"):
        hh = list(st.depth_hist); ms = list(st.depth_ms)
        print("  depth: Mrays, ms, Grays/s: " + "  ".join(f"{d}:{hh[d] / 1e6:.1f},{ms[d]:.2f},{hh[d] / ms[d] / 1e6:.1f}" for d in range(0, 12) if ms[d] > 0), flush=True)
