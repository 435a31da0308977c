"""General instance chains (VERDICT r2 "What's missing 1"): the reference's hitables compose freely
(/root/reference/RayTracer/RayTracer/Hitables.h:20-106) -- SphereHitable has a real m_center (Hitables.cpp:20,37), HitableCombo
holds any IHitable (:130-142), TranslatedInstance / RotatedInstance nest in any order (:163-173, :241-264).

CPU half: the oracle's chain-form evaluation against a literal, recursive float32 restatement of those Hit methods written here in
numpy scalars (small cases, pure Python), and against the factory form for the factories' own shapes.
GPU half: the device against the oracle, bit for bit -- per primitive, through the list, through the library's own tree in both
visiting orders, and as images."""
import math

import os

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene
from tests import helpers as H

F = np.float32
FLT_MAX = float(np.finfo(np.float32).max)


# ---- a literal restatement of the reference's recursion (float32 scalars; u,v left out: CalculateUV is covered elsewhere) ----------
def _dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))          # XMVector3Dot, SSE2 order


def _rot_tables(axis):                                                         # Hitables.cpp:188-206
    return {0: (1, 2, F(-1.0)), 1: (0, 2, F(1.0)), 2: (0, 1, F(-1.0))}[axis]


def ref_hit(node, o, d, t_min, t_max):
    """-> None or dict(t, pos, nrm, leaf) with leaf = index of the winning leaf in the tree's left-to-right order"""
    counter = [0]

    def hit(node, o, d, t_min, t_max):
        tag = node[0]
        if tag == "sphere":                                                    # Hitables.cpp:13-54
            me = counter[0]; counter[0] += 1
            c = np.asarray(node[1], F); r = F(node[2])
            oc = (o - c).astype(F)
            a = _dot(d, d); b = F(F(2.0) * _dot(d, oc)); cc = F(_dot(oc, oc) - F(r * r))
            disc = F(F(b * b) - F(F(F(4.0) * a) * cc))
            if disc > 0:
                for sgn in (-1, 1):
                    num = F(F(F(0.0) - b) - np.sqrt(disc)) if sgn < 0 else F(F(F(0.0) - b) + np.sqrt(disc))
                    t = F(num / F(F(2.0) * a))
                    if t <= t_max and t >= t_min:
                        pos = (o + (d * t).astype(F)).astype(F)
                        nrm = ((pos - c).astype(F) * F(F(1.0) / r)).astype(F)      # v / f == v * (1/f), Vec3.h:85-88
                        return dict(t=t, pos=pos, nrm=nrm, leaf=me)
            return None
        if tag == "rect":                                                      # Hitables.cpp:93-115
            me = counter[0]; counter[0] += 1
            _, aA, bA, a0, a1, b0, b1, c, rev = node
            a0, a1, b0, b1, c = F(a0), F(a1), F(b0), F(b1), F(c)
            if a0 > a1: a0, a1 = a1, a0
            if b0 > b1: b0, b1 = b1, b0
            cA = 3 - aA - bA
            with np.errstate(divide="ignore", invalid="ignore"):
                t = F(F(c - o[cA]) / d[cA])
            if t <= t_max and t >= t_min:
                a = F(o[aA] + F(t * d[aA])); b = F(o[bA] + F(t * d[bA]))
                if a >= a0 and a <= a1 and b >= b0 and b <= b1:
                    pos = (o + (d * t).astype(F)).astype(F)
                    nrm = np.zeros(3, F); nrm[cA] = F(-1.0) if rev else F(1.0)
                    return dict(t=t, pos=pos, nrm=nrm, leaf=me)
            return None
        if tag == "combo":                                                     # Hitables.cpp:130-142
            best = None
            for ch in node[1]:
                h = hit(ch, o, d, t_min, t_max)
                if h is not None:
                    t_max = h["t"]; best = h
            return best
        if tag == "translate":                                                 # Hitables.cpp:163-173
            off = np.asarray(node[2], F)
            h = hit(node[1], (o - off).astype(F), d, t_min, t_max)
            if h is not None:
                h["pos"] = (h["pos"] + off).astype(F)
            return h
        if tag == "rotate":                                                    # Hitables.cpp:182-264
            ang = F(node[2]); s, c = F(math.sin(float(ang))), F(math.cos(float(ang)))
            aA, bA, OP = _rot_tables(int(node[3])); opp = F(-OP)
            o2, d2 = o.copy(), d.copy()
            o2[aA] = F(F(c * o[aA]) + F(F(opp * s) * o[bA])); o2[bA] = F(F(F(F(-opp) * s) * o[aA]) + F(c * o[bA]))
            d2[aA] = F(F(c * d[aA]) + F(F(opp * s) * d[bA])); d2[bA] = F(F(F(F(-opp) * s) * d[aA]) + F(c * d[bA]))
            h = hit(node[1], o2, d2, t_min, t_max)
            if h is not None:
                p, n = h["pos"], h["nrm"]
                p2, n2 = p.copy(), n.copy()
                p2[aA] = F(F(c * p[aA]) + F(F(OP * s) * p[bA])); p2[bA] = F(F(F(F(-OP) * s) * p[aA]) + F(c * p[bA]))
                n2[aA] = F(F(c * n[aA]) + F(F(OP * s) * n[bA])); n2[bA] = F(F(F(F(-OP) * s) * n[aA]) + F(c * n[bA]))
                h["pos"], h["nrm"] = p2, n2
            return h
        raise ValueError(tag)

    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        return hit(node, np.asarray(o, F), np.asarray(d, F), F(t_min), F(t_max))


# ---- trees ---------------------------------------------------------------------------------------------------------------------
def verdict_example():
    """`new TranslatedInstance(new RotatedInstance(new SphereHitable(Vec3(1,2,3), r), 30, 1), ...)` (the angle is in radians,
    Hitables.cpp:184-185) and a combo of two spheres."""
    return [("translate", ("rotate", ("sphere", (1.0, 2.0, 3.0), 0.8), 30.0, 1), (0.25, -0.5, 0.125)),
            ("combo", [("sphere", (-1.5, 0.2, 0.3), 0.6), ("sphere", (-0.9, 0.5, 0.1), 0.5)])]


def random_tree(rng, depth=0):
    r = rng.random()
    if depth < 4 and r < 0.30:
        return ("translate", random_tree(rng, depth + 1), tuple(rng.uniform(-1.0, 1.0, 3)))
    if depth < 4 and r < 0.60:
        return ("rotate", random_tree(rng, depth + 1), float(rng.uniform(-3.0, 3.0)), int(rng.integers(0, 3)))
    if depth < 3 and r < 0.72:
        return ("combo", [random_tree(rng, depth + 1) for _ in range(int(rng.integers(1, 4)))])
    if r < 0.86 or depth == 0:
        return ("sphere", tuple(rng.uniform(-0.8, 0.8, 3)), float(rng.uniform(0.2, 0.9)))
    a, b = [(0, 1), (0, 2), (2, 1), (1, 0), (1, 2), (2, 0)][int(rng.integers(0, 6))]
    lo = rng.uniform(-1.0, 0.0, 2); hi = rng.uniform(0.1, 1.0, 2)
    return ("rect", a, b, float(lo[0]), float(hi[0]), float(hi[1]), float(lo[1]), float(rng.uniform(-0.5, 0.5)), bool(rng.integers(0, 2)))


def chain_scene(trees, force_chain=False, ambient=(0.6, 0.7, 0.9)):
    s = scene.FlatScene(ambient=ambient)
    mats = [s.lambertian((0.7, 0.3, 0.2)), s.metal((0.8, 0.8, 0.7), 0.1), s.dielectric(1.5), s.lambertian((0.2, 0.6, 0.3)), s.diffuse_light((3.0, 3.0, 3.0))]
    for i, t in enumerate(trees):
        s.hitable(t, mats[i % len(mats)], force_chain=force_chain)
    cam = scene.make_camera((0.5, 1.0, 7.0), (0.0, 0.3, 0.0), 45.0, 0.05, 1.0)
    return s, cam


def _rays(n, seed, extent=2.5):
    org, d = H.ray_batch(n, seed, extent=extent)
    return org, d


# ---- CPU: the oracle against the literal recursion ---------------------------------------------------------------------------------
def test_oracle_chain_form_follows_the_reference_recursion(orc):
    rng = np.random.default_rng(2024)
    trees = verdict_example() + [random_tree(rng) for _ in range(14)]
    s, _ = chain_scene(trees, force_chain=True)
    assert s.n_objects == len(trees) and any(c for c in s.chain)
    org, d = _rays(320, 5, extent=2.0)
    n_hit = 0
    for obj, tree in enumerate(trees):
        first, cnt = s.obj_first[obj], s.obj_count[obj]
        for (t0, t1) in ((0.001, FLT_MAX), (0.5, 6.0)):
            # Object::Hit = the tree's own Hit: the oracle walks the object's primitives in order with t_max shrinking
            best = np.full(org.shape[0], -1); rec = [None] * org.shape[0]; tm = np.full(org.shape[0], F(t1), F)
            for p in range(first, first + cnt):
                o = orc.prim_hit_batch(s, p, org, d, t0, t1)
                for i in range(org.shape[0]):
                    if o["prim"][i] >= 0 and o["t"][i] <= tm[i]:
                        tm[i] = o["t"][i]; best[i] = p - first; rec[i] = o[i]
            for i in range(org.shape[0]):
                want = ref_hit(tree, org[i], d[i], t0, t1)
                if want is None:
                    assert best[i] < 0, (obj, i)
                    continue
                n_hit += 1
                assert best[i] == want["leaf"], (obj, i, best[i], want["leaf"])
                assert rec[i]["t"].view(np.uint32) == want["t"].view(np.uint32), (obj, i)
                H.assert_bits(rec[i]["pos"], want["pos"], f"tree {obj} ray {i} pos")
                H.assert_bits(rec[i]["nrm"], want["nrm"], f"tree {obj} ray {i} nrm")
    assert n_hit > 250


def test_oracle_list_walk_of_chain_objects(orc):
    """world hit through the list = every object's tree in order, t_max shrinking across objects (what HitableCombo of all does)."""
    trees = verdict_example() + [random_tree(np.random.default_rng(7)) for _ in range(5)]
    s, _ = chain_scene(trees)
    org, d = _rays(300, 9, extent=2.0)
    o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, orc.FLAG_ACCEL_LIST)
    whole = ("combo", trees)
    hits = 0
    for i in range(org.shape[0]):
        want = ref_hit(whole, org[i], d[i], 0.001, FLT_MAX)
        if want is None:
            assert o["prim"][i] < 0
            continue
        hits += 1
        assert o["prim"][i] == want["leaf"] and o["t"][i].view(np.uint32) == want["t"].view(np.uint32)
        H.assert_bits(o["pos"][i], want["pos"], f"ray {i} pos"); H.assert_bits(o["nrm"][i], want["nrm"], f"ray {i} nrm")
    assert hits > 40


def test_factory_shapes_agree_in_both_forms(orc):
    """Translate(Sphere at 0) and Translate(RotZ(RotY(RotX(Rect)))) evaluated as chains give the bits of the factory form."""
    sph = ("translate", ("sphere", (0.0, 0.0, 0.0), 0.7), (0.3, -0.2, 0.5))
    rect = ("translate", ("rotate", ("rotate", ("rotate", ("rect", 0, 2, -0.8, 0.8, -0.6, 0.6, 0.0, True), 0.4, 0), -0.9, 1), 1.3, 2), (-0.4, 0.1, 0.2))
    a, _ = chain_scene([sph, rect]); b, _ = chain_scene([sph, rect], force_chain=True)
    assert all(c is None for c in a.chain) and all(c is not None for c in b.chain)
    org, d = _rays(4096, 3)
    for p in range(2):
        ha = orc.prim_hit_batch(a, p, org, d, 0.001, FLT_MAX); hb = orc.prim_hit_batch(b, p, org, d, 0.001, FLT_MAX)
        assert (ha["prim"] >= 0).sum() > 200
        H.assert_hits_equal(ha, hb, f"prim {p}: factory form vs chain form")
    ba, bb = orc.object_bounds(a), orc.object_bounds(b)
    assert np.array_equal(ba[0], bb[0]) and np.array_equal(ba[1], bb[1])       # and the same boxes


def test_chain_boxes_contain_their_hits(orc):
    rng = np.random.default_rng(11)
    trees = verdict_example() + [random_tree(rng) for _ in range(20)]
    s, _ = chain_scene(trees, force_chain=True)
    mn, mx = orc.object_bounds(s)
    org, d = _rays(2000, 13)
    for obj in range(s.n_objects):
        for p in range(s.obj_first[obj], s.obj_first[obj] + s.obj_count[obj]):
            h = orc.prim_hit_batch(s, p, org, d, 0.001, FLT_MAX)
            pos = h["pos"][h["prim"] >= 0]
            assert ((pos >= mn[obj] - 2e-4) & (pos <= mx[obj] + 2e-4)).all(), (obj, p)
    # a tree built over these boxes finds what the list finds (the winner of an exact tie aside)
    capi.build_bvh(s)
    a, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, orc.FLAG_ACCEL_LIST)
    for flags in (0, orc.FLAG_ACCEL_ORDERED):
        b, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, flags)
        H.assert_bits(a["t"], b["t"], f"tree walk flags={flags} vs list")


def test_chain_scene_validation():
    s, _ = chain_scene(verdict_example())
    with pytest.raises(ValueError, match="instance chain of 9 ops"):
        t = ("sphere", (0, 0, 0), 1.0)
        for k in range(9):
            t = ("translate", t, (0.1, 0.0, 0.0))
        s.hitable(t, 0)


# ---- GPU: the device against the oracle ---------------------------------------------------------------------------------------------
def _gpu_scene(seed, n=10):
    rng = np.random.default_rng(seed)
    trees = verdict_example() + [random_tree(rng) for _ in range(n)]
    # a floor and a light in the factory form, so that both forms live in one scene
    s, cam = chain_scene(trees)
    s.rect(scene.XZ_RECT, (0.0, -2.0, 0.0), (0.0, 0.0, 0.0), 12.0, 12.0, False, 3)
    s.rect(scene.XZ_RECT, (0.0, 5.0, 0.0), (0.0, 0.2, 0.0), 3.0, 3.0, True, 4)
    return s, cam


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 1 + int(os.environ.get("MIRT_CHAIN_SEEDS", "3"))))      # a longer campaign: MIRT_CHAIN_SEEDS=100
def test_gpu_chain_prims_and_walks(gpu, orc, seed):
    s, cam = _gpu_scene(seed)
    gpu.upload_scene(s)
    org, d = _rays(8192, 20 + seed, extent=3.0)
    for p in range(s.n_prims):
        for (t0, t1) in ((0.001, FLT_MAX), (0.25, 5.0)):
            H.assert_hits_equal(gpu.prim_hit_batch(p, org, d, t0, t1), orc.prim_hit_batch(s, p, org, d, t0, t1), f"seed {seed} prim {p}", scene=s)
    o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, orc.FLAG_ACCEL_LIST)
    assert (o["prim"] >= 0).sum() > 2000
    H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST), o, f"seed {seed} list", scene=s)
    capi.build_bvh(s)                                            # the library's own tree over the chain-form boxes, walked both ways
    gpu.upload_scene(s)
    for flags in (0, capi.FLAG_ACCEL_ORDERED):
        o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, flags)
        H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, flags), o, f"seed {seed} tree flags={flags}", scene=s)


@pytest.mark.gpu
def test_gpu_chain_images(gpu, orc):
    """The verdict's example scene renders bit-identically to the oracle: list walk, the library's tree near-first (plain and in
    resumable slices), the tree in the reference's order; and an all-factory scene is untouched by the new fields."""
    s, cam = _gpu_scene(5, n=6)
    gpu.upload_scene(s); gpu.set_camera(cam)
    w, h, spp = 96, 72, 6
    osum, ost = orc.trace(s, cam, w, h, spp, 50, 0xC0FFEE, orc.FLAG_ACCEL_LIST)
    g = gpu.render(w, h, spp, 50, 0xC0FFEE, capi.FLAG_ACCEL_LIST)
    H.assert_bits(g["sum"], osum, "chain scene, list walk"); assert g["stats"].rays == ost.rays and osum.any()
    capi.build_bvh(s); gpu.upload_scene(s)
    for flags in (0, capi.FLAG_ACCEL_ORDERED):
        osum, ost = orc.trace(s, cam, w, h, spp, 50, 0xC0FFEE, flags)
        for wy in (0, 48):
            gpu.set_option("walk_yield", wy)
            g = gpu.render(w, h, spp, 50, 0xC0FFEE, flags)
            H.assert_bits(g["sum"], osum, f"chain scene, tree flags={flags} walk_yield={wy}"); assert g["stats"].rays == ost.rays
    gpu.set_option("walk_yield", -1)
    for fl in (capi.FLAG_NORMAL_DISPLAY, capi.FLAG_ONE_SPP):
        osum, _ = orc.trace(s, cam, w, h, 1, 50, 1, orc.FLAG_ACCEL_LIST | fl)
        H.assert_bits(gpu.render(w, h, 1, 50, 1, capi.FLAG_ACCEL_LIST | fl)["sum"], osum, f"chain scene flags={fl}")


@pytest.mark.gpu
def test_gpu_chain_scene_refused_where_not_supported(gpu):
    s, cam = _gpu_scene(6, n=2)
    gpu.upload_scene(s); gpu.set_camera(cam)
    gpu.set_option("engine", 0)
    try:
        with pytest.raises(capi.MirtError, match="chain"):
            gpu.render(16, 16, 1, 5, 1, capi.FLAG_ACCEL_LIST)
    finally:
        gpu.set_option("engine", 1)
