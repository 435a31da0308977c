"""CPU tests of the oracle (oracle/rt_oracle.c): known-answer checks derived from the reference's
formulas, the committed golden vectors, and the properties the parity plan relies on."""
import json
import os

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene
from tests import helpers as H
from tests import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLT_MAX = float(np.finfo(np.float32).max)
F = np.float32


def gold(name):
    return np.load(os.path.join(GOLD, name))


def world(name):
    return {"cornell": lambda: scene.cornell_box(1.0), "two_boxes": lambda: scene.cornell_two_boxes(1.0),
            "spheres4": lambda: scene.spheres_world(2.0, small_spheres=False), "spheres": lambda: scene.spheres_world(1.5)}[name]()


# ---- RNG -------------------------------------------------------------------------------------
def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert orc.philox([[0, 0, 0, 0]], [0, 0])[0].tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox([[0xFFFFFFFF] * 4], [0xFFFFFFFF] * 2)[0].tolist() == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], [0xA4093822, 0x299F31D0])[0].tolist() == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_randomizer_mappings():
    # Randomizer.h:9-17 [min,max) and :20-26 [min,max]
    assert orc.map_minmax(0, 0.0, 1.0) == 0.0
    assert orc.map_minmax(2 ** 31, 0.0, 1.0) == 0.5
    assert orc.map_minmax(2 ** 32 - 1, 0.0, 1.0) == 1.0          # the double is < 1 but rounds to 1.0f, as in the reference
    assert orc.map_minmax(2 ** 32 - 129, 0.0, 1.0) < 1.0
    assert orc.map_minmax2(0, -1.0, 1.0) == -1.0 and orc.map_minmax2(2 ** 32 - 1, -1.0, 1.0) == 1.0
    assert abs(orc.map_minmax2(2 ** 31, -1.0, 1.0)) < 1e-9
    assert orc.map_minmax(2 ** 30, 2.0, 6.0) == 3.0


def test_division_free_mapping_is_exact_for_all_u32():
    """The device maps a word to (-1,1) as (float)fma(raw * RN(1/4294967295.0), 2.0, -1.0), without the reference's divide
    (rt_device.h map_pm1); all 2^32 inputs give the reference's float."""
    assert orc.lib().orc_check_pm1_short(0, 0xFFFFFFFF) == 0


# ---- known answers from the reference's formulas ------------------------------------------------
def test_sphere_known_answers():
    s = scene.FlatScene(); m = s.lambertian((0.5, 0.5, 0.5)); s.sphere((1.0, 2.0, 3.0), 1.0, m)
    org = [[1, 2, 8], [1, 2, 3], [1, 5, 8], [1, 2, 8], [1, 3, 8]]
    d = [[0, 0, -1], [0, 0, -2], [0, 0, -1], [0, 0, 1], [0, 0, -1]]
    h = orc.prim_hit_batch(s, 0, org, d, 0.001, FLT_MAX)
    assert h["t"][0] == 4.0 and np.array_equal(h["pos"][0], [1, 2, 4]) and np.array_equal(h["nrm"][0], [0, 0, 1])
    assert h["t"][1] == 0.5 and np.array_equal(h["nrm"][1], [0, 0, -1])      # from inside: far root, direction not normalised
    assert h["prim"][2] == -1 and h["prim"][3] == -1                            # miss, behind
    assert h["prim"][4] == -1                                                   # tangent: discriminant == 0 is a miss (Hitables.cpp:28)
    assert orc.prim_hit_batch(s, 0, org[:1], d[:1], 0.001, 3.9)["prim"][0] == -1  # t_max cuts both roots
    assert orc.prim_hit_batch(s, 0, org[:1], d[:1], 4.0, 4.0)["prim"][0] == 0     # both ends inclusive (:33)
    # CalculateUV at the +z pole of the equator: theta = atan2(1,0) = pi/2 -> u = 1 - (pi/2+pi)/(2pi) = 0.25, v = 0.5
    assert abs(h["u"][0] - 0.25) < 1e-6 and abs(h["v"][0] - 0.5) < 1e-6


def test_sphere_uv_follows_calculateuv_to_a_few_ulp():
    """CalculateUV (Hitables.cpp:62-78) calls the C runtime's atan2f/asinf; the oracle and the device share one stated float
    sequence for them instead (det_atan2f / det_asinf).  That sequence must stay a faithful atan2/asin: against the float64
    formula on the same hit points, u and v differ by a few float ulp at most, over the whole sphere incl. poles and seams."""
    s = scene.FlatScene(); m = s.lambertian((0.5, 0.5, 0.5)); s.sphere((0.0, 0.0, 0.0), 1.0, m)
    rng = np.random.default_rng(5)
    n = 200000
    tgt = rng.normal(size=(n, 3)); tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    tgt[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    tgt[6:1000, 2] *= 1e-4                                                      # crowd the u seam (z ~ 0, x < 0 and > 0)
    tgt[1000:2000, [0, 2]] *= 1e-3                                              # and the poles
    org = (tgt * 3.0).astype(F); d = (-tgt).astype(F)
    h = orc.prim_hit_batch(s, 0, org, d, 0.001, FLT_MAX)
    ok = h["prim"] == 0
    assert ok.sum() > n - 10
    pn = (h["pos"][ok] / np.sqrt((h["pos"][ok] ** 2).sum(axis=1, dtype=F))[:, None].astype(F)).astype(np.float64)
    u = 1.0 - (np.arctan2(pn[:, 2], pn[:, 0]) + np.pi) / (2 * np.pi)
    v = (np.arcsin(np.clip(pn[:, 1], -1, 1)) + np.pi / 2) / np.pi
    assert np.abs(h["u"][ok] - u).max() < 4 * 2.0 ** -24 and np.abs(h["v"][ok] - v).max() < 4 * 2.0 ** -24
    assert h["u"][ok].min() >= 0.0 and h["u"][ok].max() <= 1.0 and h["v"][ok].min() >= 0.0 and h["v"][ok].max() <= 1.0


def test_rect_known_answers():
    s = scene.FlatScene(); m = s.lambertian((0.5, 0.5, 0.5))
    s.rect(scene.XZ_RECT, (0.0, 1.0, 0.0), (0, 0, 0), 2.0, 4.0, True, m)          # a = x in [-1,1], b = z in [-2,2], y = 1, normal -y
    org = [[0.5, 0, 1], [1, 0, 2], [1.0001, 0, 0], [0, 2, 0], [0, 1, 0]]
    d = [[0, 1, 0], [0, 2, 0], [0, 1, 0], [0, 1, 0], [1, 0, 0]]
    h = orc.prim_hit_batch(s, 0, org, d, 0.001, FLT_MAX)
    assert h["t"][0] == 1.0 and np.array_equal(h["pos"][0], [0.5, 1, 1]) and np.array_equal(h["nrm"][0], [0, -1, 0])
    assert h["u"][0] == 0.75 and h["v"][0] == 0.75
    assert h["t"][1] == 0.5 and h["u"][1] == 1.0 and h["v"][1] == 1.0            # edges are inside (Hitables.cpp:100)
    assert h["prim"][2] == -1 and h["prim"][3] == -1 and h["prim"][4] == -1      # outside, behind, parallel (0/0 = NaN)


def test_aabb_known_answers():
    mn, mx = [[-1, -1, -1]], [[1, 1, 1]]
    org = [[0, 0, 5], [0, 0, 5], [2, 0, 5], [0, 0, 0], [1, 0, 5]]
    d = [[0, 0, -1], [0, 0, 1], [0, 0, -1], [1, 1, 1], [0, 0, -1]]
    assert orc.aabb_hit_batch(mn, mx, org, d, 0.001, FLT_MAX)[:, 0].tolist() == [1, 0, 0, 1, 1]
    assert orc.aabb_hit_batch(mn, mx, org[:1], d[:1], 0.001, 4.0)[0, 0] == 0      # t_max <= entry (AABB.cpp:17)
    assert orc.aabb_hit_batch(mn, mx, org[:1], d[:1], 0.001, 4.5)[0, 0] == 1


def test_scatter_known_answers():
    s, _ = world("cornell")
    hits = np.zeros(4, capi.HIT_DTYPE)
    hits["nrm"] = [0, 1, 0]; hits["pos"] = [1, 2, 3]
    hits["mat"] = [0, 3, 5, 5]                                   # light, white lambertian, glass, glass
    d = np.array([[0, -1, 0], [0, -1, 0], [0, -2, 0], [1, -0.001, 0]], np.float32)
    draws = np.zeros((4, 12), np.uint32)
    draws[:, :3] = 2 ** 31                                        # first candidate ~ (0,0,0): accepted
    draws[3, 0] = 2 ** 32 - 1                                     # glass #2: unorm = 1.0 -> never < reflect_prob < 1 -> refract
    o = orc.scatter_batch(s, hits, d, draws)
    assert o["alive"].tolist() == [0, 1, 1, 1]
    assert np.array_equal(o["emit"][0], [15, 15, 15]) and o["consumed"][0] == 0
    assert np.allclose(o["dir"][1], [0, 1, 0], atol=1e-6) and np.array_equal(o["atten"][1], np.float32([0.73, 0.73, 0.73]))
    assert o["consumed"].tolist() == [0, 3, 1, 1]
    assert np.array_equal(o["atten"][2], [1, 1, 1])
    # straight down into glass with unorm 0.5 >= schlick(1, 1.5) = 0.04: refracted, undeviated: eta*(v - n*dt) - n*sqrt(disc) = (0,-1,0)
    assert np.allclose(o["dir"][2], [0, -1, 0], atol=1e-6)
    light_back = orc.scatter_batch(s, hits[0:1], [[0, 1, 0]], draws[0:1])                  # a light emits from both faces (Materials.cpp:260-264)
    assert light_back["alive"][0] == 0 and np.array_equal(light_back["emit"][0], [15, 15, 15])
    lam_back = orc.scatter_batch(s, hits[1:2], [[0, 1, 0]], draws[1:2])
    assert lam_back["alive"][0] == 0 and lam_back["consumed"][0] == 3   # one-sided (Materials.cpp:32), draws still consumed


def test_camera_matches_python_mirror():
    for (lf, la, fov, ap, asp) in (((0.0, 0.64, 4.63), (0.0, -0.26, -1.0), 30.0, 0.0, 1.0), ((11.0, 2.0, 3.0), (0.0, 0.6, 0.0), 20.0, 0.1, 1.5)):
        a = np.frombuffer(bytes(orc.camera_init(lf, la, fov, ap, asp)), np.float32)
        b = np.frombuffer(bytes(scene.make_camera(lf, la, fov, ap, asp)), np.float32)
        assert np.allclose(a, b, rtol=2e-7, atol=1e-7)
    c = orc.camera_init((0, 0, 0), (0, 0, -1), 90.0, 0.0, 2.0)
    assert np.allclose(list(c.top_left), [-2, 1, -1], atol=1e-6) and np.allclose(list(c.vertical), [0, -2, 0], atol=1e-6)   # row 0 is the top


# ---- golden vectors -------------------------------------------------------------------------------
def test_golden_prim_hits():
    g = gold("f1_prim_hits.npz")
    for key in [k for k in g.files if k.endswith("_hit") and not k.startswith("aabb")]:
        wn, p = key.rsplit("_", 2)[0], int(key.rsplit("_", 2)[1])
        s, _ = world(wn)
        o = orc.prim_hit_batch(s, p, g[f"{wn}_{p}_org"], g[f"{wn}_{p}_dir"], 0.001, FLT_MAX)
        assert o.tobytes() == g[key].tobytes(), key
        o4 = orc.prim_hit_batch(s, p, g[f"{wn}_{p}_org"], g[f"{wn}_{p}_dir"], 0.001, 4.0)
        assert o4.tobytes() == g[key + "4"].tobytes()
    assert np.array_equal(orc.aabb_hit_batch(g["aabb_min"], g["aabb_max"], g["aabb_org"], g["aabb_dir"], 0.001, FLT_MAX), g["aabb_hit"])
    assert np.array_equal(orc.aabb_hit_batch(g["aabb_min"], g["aabb_max"], g["aabb_org"], g["aabb_dir"], 0.001, 3.0), g["aabb_hit3"])


def test_golden_world_hits_and_reference_bvh():
    g = gold("f2_world_hits.npz")
    for wn in ("cornell", "two_boxes", "spheres4", "spheres"):
        s, _ = world(wn)
        n = orc.bvh_build_reference(s, g[f"{wn}_axis_draws"])
        assert n == g[f"{wn}_left"].shape[0] and np.array_equal(s.nodes[2], g[f"{wn}_left"]) and np.array_equal(s.nodes[0], g[f"{wn}_node_min"])
        b, _ = orc.world_hit_batch(s, g[f"{wn}_org"], g[f"{wn}_dir"], 0.001, FLT_MAX, 0)
        l, _ = orc.world_hit_batch(s, g[f"{wn}_org"], g[f"{wn}_dir"], 0.001, FLT_MAX, orc.FLAG_ACCEL_LIST)
        assert b.tobytes() == g[f"{wn}_bvh"].tobytes() and l.tobytes() == g[f"{wn}_list"].tobytes()
        H.assert_bits(b["t"], l["t"], f"{wn}: BVH and list closest t")           # any valid BVH finds the same closest t


def test_golden_scatter_and_camera():
    g = gold("f3_scatter.npz")
    s, _ = world("cornell"); s.dielectric(2.4); s.metal((0.7, 0.6, 0.5), 0.0); s.metal((0.9, 0.9, 0.9), 5.0)
    o = orc.scatter_batch(s, g["hits"], g["dir_in"], g["draws"])
    for k, v in o.items():
        assert v.tobytes() == g["out_" + k].tobytes(), k
    c = gold("f4_camera_rays.npz")
    for name in ("cornell", "spheres", "spheres4"):
        cam = capi.CameraStruct.from_buffer_copy(c[f"{name}_cam"].tobytes())
        o = orc.camera_rays(cam, c[f"{name}_uv"], c[f"{name}_draws"])
        assert o["org"].tobytes() == c[f"{name}_org"].tobytes() and o["dir"].tobytes() == c[f"{name}_dir"].tobytes()
        assert np.array_equal(o["consumed"], c[f"{name}_consumed"])


def test_golden_images():
    g = gold("f5_images.npz")
    for wn, (w, h) in {"cornell": (128, 128), "spheres4": (200, 100), "two_boxes": (64, 64)}.items():
        s, cam = world(wn)
        orc.bvh_build_reference(s, g[f"{wn}_axis_draws"])
        for tag, flags, spp in (("1spp", orc.FLAG_ONE_SPP, 1), ("normal", orc.FLAG_NORMAL_DISPLAY | orc.FLAG_ONE_SPP, 1), ("8spp", 0, 8)):
            ssum, _ = orc.trace(s, cam, w, h, spp, 50, 0x5EED, flags)
            _, rg = orc.resolve(ssum, spp, flags)
            assert np.array_equal(rg, g[f"{wn}_{tag}_rgba8"]), (wn, tag)
        assert ssum.tobytes() == g[f"{wn}_8spp_sum"].tobytes()


# ---- properties ---------------------------------------------------------------------------------
def test_render_loop_properties():
    s, cam = world("cornell")
    a, st = orc.trace(s, cam, 48, 48, 6, 50, 11, orc.FLAG_ACCEL_LIST, threads=1)
    b, _ = orc.trace(s, cam, 48, 48, 6, 50, 11, orc.FLAG_ACCEL_LIST, threads=4)
    assert a.tobytes() == b.tobytes()                                        # thread count does not matter
    c, _ = orc.trace(s, cam, 48, 48, 2, 50, 11, orc.FLAG_ACCEL_LIST)
    c, _ = orc.trace(s, cam, 48, 48, 4, 50, 11, orc.FLAG_ACCEL_LIST, spp_begin=2, sum_rgb=c)
    assert a.tobytes() == c.tobytes()                                        # sample ranges compose (ordered sum)
    assert 5.5 < st.rays / (48 * 48 * 6) < 7.5                               # 6.57 rays per sample
    r, _ = orc.trace(s, cam, 48, 48, 6, 50, 11, orc.FLAG_ACCEL_LIST | orc.FLAG_RECURSIVE)
    # forward (throughput) order vs the reference's recursive order: float re-association only
    assert np.allclose(a, r, rtol=3e-6, atol=1e-7)
    z, zs = orc.trace(s, cam, 32, 32, 2, 0, 11, orc.FLAG_ACCEL_LIST)
    assert not z.any() and zs.rays == 32 * 32 * 2                            # depth 0: Scatter never called, even a light returns 0
    e, _ = orc.trace(s, cam, 16, 16, 1, 50, 1, orc.FLAG_ACCEL_LIST, rng=orc.RNG_ENTROPY, threads=1)
    assert np.isfinite(e).all()


def test_resolve_and_ppm(tmp_path):
    ssum = np.array([[[0.0, 4.0, 1.0], [-1.0, 0.25 * 4, 9.0]]], np.float32)    # 1x2 image, spp 4 -> means 0,1,.25 / -.25,.25,2.25
    px, rg = orc.resolve(ssum, 4)
    assert np.allclose(px[0, 0], [0, 1, 0.5]) and np.isnan(px[0, 1, 0])       # sqrt of a negative mean is NaN -> byte 0
    assert rg[0, 0].tolist() == [0, 255, 127, 255] and rg[0, 1].tolist() == [0, 127, 255, 255]   # (UINT32)(c*255.0f), truncation
    px1, rg1 = orc.resolve(ssum, 4, orc.FLAG_ONE_SPP)                          # 1-SPP: no divide (HomemadeRayTracer.cpp:120)
    assert np.allclose(px1[0, 0], [0, 2, 1])
    p = tmp_path / "o.ppm"
    assert orc.write_ppm(str(p), rg) == 0
    assert p.read_text() == "P3\n2 1\n255\n0 255 127\n0 127 255\n"


def test_converged_images_match_across_seeds():
    side = json.load(open(os.path.join(GOLD, "f6_converged.json")))
    g = gold("f6_converged.npz")
    for wn, meta in side.items():
        a, b = g[f"{wn}_seed101"].astype(np.float32), g[f"{wn}_seed202"].astype(np.float32)
        assert abs(H.psnr(a, b) - meta["mutual_psnr_db"]) < 0.05
        assert meta["mutual_psnr_db"] > 25.0


def test_object_bounds_and_bvh_shape():
    s, _ = world("two_boxes")
    mn, mx = orc.object_bounds(s)
    assert np.allclose(mn[3], [-1, -1, -1.0001], atol=1e-6) and np.allclose(mx[3], [1, 1, -0.9999], atol=1e-6)   # back wall, padded in z
    assert mn[6][1] == pytest.approx(-1.0001, abs=1e-6) and mx[6][0] - mn[6][0] > 0.6          # rotated cube: wider than its side
    n = orc.bvh_build_reference(s, np.zeros(64, np.uint32))                                     # all draws 0 -> axis x everywhere
    assert n == 7 and s.nodes[4] == 0


def test_random_scenes_bvh_equals_list():
    """Random mixtures of every primitive kind: the reference-shaped tree and the plain list give the same closest t."""
    from tests.test_gpu_parity import random_scene
    for seed in range(6):
        s, cam = random_scene(100 + seed)
        draws = np.random.default_rng(seed).integers(0, 2 ** 32, 4 * s.n_objects + 8, dtype=np.uint64).astype(np.uint32)
        orc.bvh_build_reference(s, draws)
        org, d = H.ray_batch(4096, 40 + seed, extent=5.0)
        a, _ = orc.world_hit_batch(s, org, d, 0.001, float(np.finfo(np.float32).max), 0)
        b, _ = orc.world_hit_batch(s, org, d, 0.001, float(np.finfo(np.float32).max), orc.FLAG_ACCEL_LIST)
        assert (a["prim"] >= 0).sum() > 300
        H.assert_bits(a["t"], b["t"], f"seed {seed}")
        img, st = orc.trace(s, cam, 24, 16, 2, 50, seed, 0)
        assert np.isfinite(img).all() and st.rays >= 24 * 16 * 2
