#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the CPU oracle (oracle/rt_oracle.c).

The reference holds no tests or golden vectors for this path and cannot be
built here, so these vectors pin the *oracle* (and, on the GPU box, the HIP path) against
regressions; they are not outputs of the reference itself -- parity stays "unpinned" in that sense.
Inputs are seeded and stored with the outputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracinginaweekend_amd import capi, scene  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
FLT_MAX = float(np.finfo(np.float32).max)


def world(name):
    return {"cornell": lambda: scene.cornell_box(1.0), "two_boxes": lambda: scene.cornell_two_boxes(1.0),
            "spheres4": lambda: scene.spheres_world(2.0, small_spheres=False), "spheres": lambda: scene.spheres_world(1.5)}[name]()


def attach_tree(s, seed):
    draws = np.random.default_rng(seed).integers(0, 2 ** 32, 4 * s.n_objects + 8, dtype=np.uint64).astype(np.uint32)
    orc.bvh_build_reference(s, draws)
    return draws


def make_f7():
    """F7: image textures -- texel fetches (edge cases included), textured Scatter with explicit draws, a small image."""
    soil, checker = H.synthetic_texture(37, 23, 1, "ramp"), H.synthetic_texture(64, 48, 2, "checker")
    s, cam = scene.textured_world(1.5, soil, checker)
    rng = np.random.default_rng(17)
    uv = rng.uniform(-0.05, 1.05, (4096, 2)).astype(np.float32)
    uv[:14] = np.array([[0, 0], [1, 1], [0, 1], [1, 0], [0.999999, 1e-7], [-0.25, 0.5], [1.5, -0.5], [0.5, 2.0], [-3.0, 0.5],
                        [np.nan, 0.5], [0.5, np.nan], [np.inf, 0.5], [0.5, -np.inf], [1e30, 1e-30]], np.float32)
    f7 = dict(soil=soil, checker=checker, uv=uv, sample0=orc.tex_sample_batch(s, 0, uv), sample1=orc.tex_sample_batch(s, 1, uv))
    n = 2048
    hits = np.zeros(n, capi.HIT_DTYPE)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits["nrm"] = nrm.astype(np.float32); hits["pos"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32); hits["t"] = 1.0
    hits["mat"] = np.arange(n) % len(s.mkind); hits["u"] = uv[:n, 0]; hits["v"] = uv[:n, 1]
    d = rng.normal(size=(n, 3)).astype(np.float32)
    draws = rng.integers(0, 2 ** 32, (n, 3 * 64), dtype=np.uint64).astype(np.uint32)
    o = orc.scatter_batch(s, hits, d, draws)
    keep = int(o["consumed"].max())
    f7.update(hits=hits, dir_in=d, draws=draws[:, :keep], **{"out_" + k: v for k, v in o.items()})
    f7["image_sum"], _ = orc.trace(s, cam, 60, 40, 4, 50, 3, orc.FLAG_ACCEL_LIST)
    np.savez_compressed(os.path.join(OUT, "f7_textures.npz"), **f7)


def main():
    if sys.argv[1:] == ["f7"]:
        make_f7()
        return
    make_f7()
    # F1: per-primitive hits + AABB
    f1 = {}
    for wn in ("cornell", "two_boxes", "spheres4"):
        s, _ = world(wn)
        for p in ([0, 2, 3, 6, 7] if wn == "cornell" else ([6, 9, 16] if wn == "two_boxes" else [0, 2])):
            gm = s.geom[p]
            ext = 0.6 * max(gm[1] - gm[0], gm[3] - gm[2]) if s.kind[p] == capi.PRIM_RECT else 2.0 * gm[0]
            org, d = H.ray_batch(1024, 1000 + p, center=s.arrays()["prim_off"][p], extent=min(ext, 50.0))
            f1[f"{wn}_{p}_org"] = org; f1[f"{wn}_{p}_dir"] = d
            f1[f"{wn}_{p}_hit"] = orc.prim_hit_batch(s, p, org, d, 0.001, FLT_MAX)
            f1[f"{wn}_{p}_hit4"] = orc.prim_hit_batch(s, p, org, d, 0.001, 4.0)
    rng = np.random.default_rng(3)
    mn = rng.uniform(-2, 0, (16, 3)).astype(np.float32); mx = (mn + rng.uniform(0, 2, (16, 3))).astype(np.float32)
    org, d = H.ray_batch(2048, 4)
    f1.update(aabb_min=mn, aabb_max=mx, aabb_org=org, aabb_dir=d, aabb_hit=orc.aabb_hit_batch(mn, mx, org, d, 0.001, FLT_MAX),
              aabb_hit3=orc.aabb_hit_batch(mn, mx, org, d, 0.001, 3.0))
    np.savez_compressed(os.path.join(OUT, "f1_prim_hits.npz"), **f1)

    # F2: world hits through the reference-shaped BVH and the list
    f2 = {}
    for wn in ("cornell", "two_boxes", "spheres4", "spheres"):
        s, _ = world(wn)
        f2[f"{wn}_axis_draws"] = attach_tree(s, 7)
        mnn, mxx, l, r, root = s.nodes
        f2[f"{wn}_node_min"] = mnn; f2[f"{wn}_node_max"] = mxx; f2[f"{wn}_left"] = l; f2[f"{wn}_right"] = r; f2[f"{wn}_root"] = np.int32(root)
        org, d = H.ray_batch(4096, 11, extent=2.0 if wn in ("cornell", "two_boxes") else 12.0)
        f2[f"{wn}_org"] = org; f2[f"{wn}_dir"] = d
        f2[f"{wn}_bvh"], _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, 0)
        f2[f"{wn}_list"], _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, orc.FLAG_ACCEL_LIST)
    np.savez_compressed(os.path.join(OUT, "f2_world_hits.npz"), **f2)

    # F3: scatter with explicit draws (materials: cornell's 7 + glass 2.4, mirror, rough metal)
    s, _ = world("cornell")
    s.dielectric(2.4); s.metal((0.7, 0.6, 0.5), 0.0); s.metal((0.9, 0.9, 0.9), 5.0)
    rng = np.random.default_rng(5)
    n = 256 * len(s.mkind)
    hits = np.zeros(n, capi.HIT_DTYPE)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits["nrm"] = nrm.astype(np.float32); hits["pos"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    hits["t"] = 1.0; hits["mat"] = np.arange(n) % len(s.mkind)
    d = (rng.normal(size=(n, 3)) * 3.0).astype(np.float32)
    graze = rng.random(n) < 0.2
    d[graze] = (np.cross(nrm[graze], rng.normal(size=(graze.sum(), 3))) + 0.05 * nrm[graze] * rng.normal(size=(graze.sum(), 1))).astype(np.float32)
    d[0] = 0.0
    draws = rng.integers(0, 2 ** 32, (n, 3 * 64), dtype=np.uint64).astype(np.uint32)
    o = orc.scatter_batch(s, hits, d, draws)
    keep = int(o["consumed"].max())
    np.savez_compressed(os.path.join(OUT, "f3_scatter.npz"), hits=hits, dir_in=d, draws=draws[:, :keep], **{"out_" + k: v for k, v in o.items()})

    # F4: camera rays
    f4 = {}
    rng = np.random.default_rng(6)
    for name, (mk, aspect) in {"cornell": (scene.cornell_box, 1.0), "spheres": (scene.spheres_world, 1.5),
                               "spheres4": (lambda a: scene.spheres_world(a, small_spheres=False), 2.0)}.items():
        _, cam = mk(aspect)
        uv = rng.uniform(0, 1, (1024, 2)).astype(np.float32)
        dr = rng.integers(0, 2 ** 32, (1024, 128), dtype=np.uint64).astype(np.uint32)
        o = orc.camera_rays(cam, uv, dr)
        keep = int(o["consumed"].max())
        f4[f"{name}_cam"] = np.frombuffer(bytes(cam), np.float32).copy(); f4[f"{name}_uv"] = uv; f4[f"{name}_draws"] = dr[:, :keep]
        f4[f"{name}_org"] = o["org"]; f4[f"{name}_dir"] = o["dir"]; f4[f"{name}_consumed"] = o["consumed"]
    np.savez_compressed(os.path.join(OUT, "f4_camera_rays.npz"), **f4)

    # F5: deterministic images (RGBA8) and a short multi-sample float image, seed 0x5EED, BVH from seed 9
    f5 = {}
    for wn, (w, h) in {"cornell": (128, 128), "spheres4": (200, 100), "two_boxes": (64, 64)}.items():
        s, cam = world(wn)
        f5[f"{wn}_axis_draws"] = attach_tree(s, 9)
        for tag, flags, spp in (("1spp", orc.FLAG_ONE_SPP, 1), ("normal", orc.FLAG_NORMAL_DISPLAY | orc.FLAG_ONE_SPP, 1), ("8spp", 0, 8)):
            ssum, _ = orc.trace(s, cam, w, h, spp, 50, 0x5EED, flags)
            _, rg = orc.resolve(ssum, spp, flags)
            f5[f"{wn}_{tag}_rgba8"] = rg
            if tag == "8spp":
                f5[f"{wn}_{tag}_sum"] = ssum
    np.savez_compressed(os.path.join(OUT, "f5_images.npz"), **f5)

    # F6: converged images from two independent seeds + their mutual PSNR (the noise floor)
    side = {}
    f6 = {}
    for wn, (w, h, spp) in {"cornell": (96, 96, 4096), "spheres4": (100, 50, 2048)}.items():
        s, cam = world(wn)
        imgs = []
        for seed in (101, 202):
            ssum, _ = orc.trace(s, cam, w, h, spp, 50, seed, orc.FLAG_ACCEL_LIST | orc.FLAG_RECURSIVE)
            px, _ = orc.resolve(ssum, spp)
            imgs.append(np.clip(px, 0, 1))
        f6[f"{wn}_seed101"] = imgs[0].astype(np.float16); f6[f"{wn}_seed202"] = imgs[1].astype(np.float16)
        side[wn] = {"width": w, "height": h, "spp": spp, "seeds": [101, 202], "order": "recursive (reference order)",
                    "mutual_psnr_db": H.psnr(imgs[0], imgs[1])}
    np.savez_compressed(os.path.join(OUT, "f6_converged.npz"), **f6)
    json.dump(side, open(os.path.join(OUT, "f6_converged.json"), "w"), indent=1)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
