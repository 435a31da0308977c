"""GPU parity: the HIP path through the C ABI against the CPU oracle, on the same seeded inputs.
Bar: bit-exact (sign of zero excepted where rt_device.h skips identity rotations), sphere u,v included: atan2f/asinf
are one float sequence shared with the oracle (det_atan2f / det_asinf)."""
import os

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene
from tests import helpers as H

pytestmark = pytest.mark.gpu

FLT_MAX = float(np.finfo(np.float32).max)


def test_arith_blocks(gpu, orc):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(size=20000) * 10.0 ** rng.uniform(-20, 20, 20000), [0.0, -0.0, 1e-40, 3e38, np.inf]]).astype(np.float32)
    y = np.concatenate([rng.normal(size=20000) * 10.0 ** rng.uniform(-20, 20, 20000), [0.0, 1.0, 1e-40, 3e38, np.inf]]).astype(np.float32)
    with np.errstate(all="ignore"):
        H.assert_bits(gpu.arith_batch(0, x, y), (x / y).astype(np.float32), "div")
        H.assert_bits(gpu.arith_batch(1, np.abs(x)), np.sqrt(np.abs(x)).astype(np.float32), "sqrt")
        p = rng.uniform(-1.5, 1.5, 20000).astype(np.float32)
        H.assert_bits(gpu.arith_batch(2, p), (p.astype(np.float64) ** 2 * p.astype(np.float64) ** 2 * p.astype(np.float64)).astype(np.float32), "pow5")
    # the words where a shortened (-1,1) mapping would go wrong first -- both ends and the zero crossing at 2^31 (rt_device.h map_pm1;
    # the CPU test walks all 2^32 words through the same formula) -- then a random million, then the six edge words the slices below expect last
    dense = np.concatenate([np.arange(0, 8192, dtype=np.uint64), np.arange(2 ** 31 - 8192, 2 ** 31 + 8192, dtype=np.uint64), np.arange(2 ** 32 - 8192, 2 ** 32, dtype=np.uint64)])
    raw = np.concatenate([rng.integers(0, 2 ** 32, 50000, dtype=np.uint64), dense, rng.integers(0, 2 ** 32, 1000000, dtype=np.uint64),
                          [0, 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - 128, 2 ** 32 - 129]]).astype(np.uint32)
    exp0 = np.array([orc.map_minmax(r, 0.0, 1.0) for r in raw[:4000]] + [orc.map_minmax(r, 0.0, 1.0) for r in raw[-6:]], np.float32)
    exp1 = np.array([orc.map_minmax2(r, -1.0, 1.0) for r in raw[:4000]] + [orc.map_minmax2(r, -1.0, 1.0) for r in raw[-6:]], np.float32)
    sel = np.concatenate([raw[:4000], raw[-6:]])
    H.assert_bits(gpu.arith_batch(3, sel.view(np.float32)), exp0, "map_unorm")
    H.assert_bits(gpu.arith_batch(4, sel.view(np.float32)), exp1, "map_pm1")
    # all raw values against the closed forms, vectorised
    d = raw.astype(np.float64)
    H.assert_bits(gpu.arith_batch(3, raw.view(np.float32)), (d / 4294967296.0).astype(np.float32), "map_unorm/all")
    H.assert_bits(gpu.arith_batch(4, raw.view(np.float32)), ((d / 4294967295.0) * 2.0 + -1.0).astype(np.float32), "map_pm1/all")


def test_lean_square_root(gpu):
    """sqrt_in_range / sqrt_wave (rt_device.h): the render kernels' square root must be the correctly rounded one.  Every
    float of eight whole binades spread over [2^-60, 2^60] (2^26 operands: all mantissas, even and odd exponents, both ends
    of the range), the range limits, and a mixed batch with zeros, denormals, infinities, negatives and NaN in the same
    waves (the general path) -- against numpy's IEEE sqrt, bit for bit."""
    mant = np.arange(1 << 23, dtype=np.uint32)
    for e in (-60, -59, -31, -2, -1, 0, 1, 30, 59):
        x = (np.uint32((127 + e) << 23) | mant).view(np.float32)
        want = np.sqrt(x.astype(np.float64)).astype(np.float32)                  # double sqrt rounded to float = correctly rounded float sqrt
        H.assert_bits(gpu.arith_batch(8, x), want, f"sqrt_in_range, binade 2^{e}")
        H.assert_bits(gpu.arith_batch(7, x[::7]), want[::7], f"sqrt_wave, binade 2^{e}")
    edge = np.array([2.0 ** -60, 2.0 ** 60, np.nextafter(np.float32(2.0 ** 60), np.float32(0))], np.float32)
    H.assert_bits(gpu.arith_batch(8, edge), np.sqrt(edge.astype(np.float64)).astype(np.float32), "range limits")
    rng = np.random.default_rng(3)
    mix = (rng.normal(size=65536) * 10.0 ** rng.uniform(-44, 38, 65536)).astype(np.float32)
    mix[::97] = 0.0; mix[5::101] = np.inf; mix[7::103] = np.nan; mix[11::89] = 1e-41; mix[13::83] = -0.0
    with np.errstate(all="ignore"):
        H.assert_bits(gpu.arith_batch(7, mix), np.sqrt(mix.astype(np.float64)).astype(np.float32), "sqrt_wave, mixed waves")


def test_reciprocal_integer_division(gpu):
    """fast_div (rt_device.h): path id / pixels-of-a-batch and pixel / width through a host-made 33-bit reciprocal must be the
    integer quotient for every 32-bit numerator: all divisors up to 4100 and a spread of large ones (powers of two, their
    neighbours, primes, 2^32 - 1) against numerators at both ends of the range, around every multiple boundary, and random."""
    rng = np.random.default_rng(11)
    divs = np.concatenate([np.arange(1, 4101, dtype=np.uint64), 2 ** np.arange(12, 32, dtype=np.uint64), 2 ** np.arange(12, 32, dtype=np.uint64) - 1,
                           2 ** np.arange(12, 32, dtype=np.uint64) + 1, np.array([1048573, 2073600, 16777259, 2 ** 31 + 11, 2 ** 32 - 2, 2 ** 32 - 1], np.uint64),
                           rng.integers(4100, 2 ** 32, 3000, dtype=np.uint64)])
    for chunk in np.array_split(divs, 8):
        k = rng.integers(0, 2 ** 32, (chunk.size, 24), dtype=np.uint64) // chunk[:, None]         # multiples k*d near which the quotient steps
        near = (k * chunk[:, None])[:, :, None] + np.array([-1, 0, 1], np.int64)[None, None, :].astype(np.uint64)
        n = np.concatenate([near.reshape(chunk.size, -1) & np.uint64(0xFFFFFFFF), rng.integers(0, 2 ** 32, (chunk.size, 52), dtype=np.uint64),
                            np.tile(np.array([0, 1, 2, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 2, 2 ** 32 - 1], np.uint64), (chunk.size, 1))], axis=1)
        d = np.repeat(chunk[:, None], n.shape[1], axis=1)
        got = gpu.arith_batch(9, n.astype(np.uint32).ravel().view(np.float32), d.astype(np.uint32).ravel().view(np.float32)).view(np.uint32)
        assert np.array_equal(got, (n // d).astype(np.uint32).ravel())


def test_shared_denominator_division(gpu):
    """div_prepare/div_by (rt_device.h: the hardware division sequence with the denominator-only part hoisted) must
    give the bits of `/` wherever it promises them (|den| in [2^-20, 2^20], |x| in [2^-40, 2^40] or x == +0): 2^26
    random pairs reaching a little beyond that range, and the range limits.  Below the numerator range it only
    promises a quotient under 2^-19 in magnitude."""
    rng = np.random.default_rng(2024)
    n = 1 << 22
    for rep in range(16):
        mant = rng.integers(0, 1 << 23, (2, n), dtype=np.uint32)
        sign = rng.integers(0, 2, (2, n), dtype=np.uint32) << 31
        ex = rng.integers(127 - 44, 127 + 43, n, dtype=np.uint32)                # numerators a little beyond [2^-40, 2^40]
        ey = rng.integers(127 - 22, 127 + 23, n, dtype=np.uint32)                # denominators a little beyond [2^-20, 2^20]
        x = (sign[0] | (ex << 23) | mant[0]).view(np.float32)
        y = (sign[1] | (ey << 23) | mant[1]).view(np.float32)
        if rep == 0:
            x[:4096] = 0.0
            y[:1024] = np.float32(3.0); x[8192:12288] = np.linspace(-2, 2, 4096, dtype=np.float32)
            x[12288:12800] = np.float32(2.0 ** -40); x[12800:13312] = np.float32(2.0 ** 40)
            y[13312:13824] = np.float32(2.0 ** -20); y[13824:14336] = np.float32(-2.0 ** 20)
        with np.errstate(all="ignore"):
            want = (x / y).astype(np.float32)
        got = gpu.arith_batch(5, x, y)
        inside = gpu.arith_batch(6, x, y) == 1.0
        ref_inside = (((np.abs(x) >= 2.0 ** -40) & (np.abs(x) <= 2.0 ** 40)) | (x.view(np.uint32) == 0)) & (np.abs(y) >= 2.0 ** -20) & (np.abs(y) <= 2.0 ** 20)
        assert np.array_equal(inside, ref_inside) and inside.mean() > 0.8
        assert np.array_equal(got.view(np.uint32)[inside], want.view(np.uint32)[inside]), f"block {rep}"     # the sign of a zero included
        small = (np.abs(x) < 2.0 ** -40) & (np.abs(y) >= 2.0 ** -20) & (np.abs(y) <= 2.0 ** 20)
        assert small.any() and (np.abs(got[small]) < 2.0 ** -19).all()


def test_philox(gpu, orc):
    rng = np.random.default_rng(2)
    ctr = rng.integers(0, 2 ** 32, (512, 4), dtype=np.uint64).astype(np.uint32)
    ctr[0] = 0; ctr[1] = 0xFFFFFFFF
    key = np.array([0xA4093822, 0x299F31D0], np.uint32)
    assert np.array_equal(gpu.philox_batch(ctr, key), orc.philox(ctr, key))
    kat = gpu.philox_batch([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], key)
    assert kat[0].tolist() == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]      # Random123 kat_vectors


def test_aabb(gpu, orc):
    rng = np.random.default_rng(3)
    mn = rng.uniform(-2, 0, (32, 3)).astype(np.float32); mx = (mn + rng.uniform(0, 2, (32, 3))).astype(np.float32)
    mx[0] = mn[0]                                                  # degenerate box
    org, d = H.ray_batch(8192, 4)
    for (t0, t1) in ((0.001, FLT_MAX), (0.001, 3.0), (1.5, 2.5)):
        assert np.array_equal(gpu.aabb_hit_batch(mn, mx, org, d, t0, t1), orc.aabb_hit_batch(mn, mx, org, d, t0, t1))


@pytest.mark.parametrize("world", ["cornell", "two_boxes", "spheres4"])
def test_prim_hit(gpu, orc, world):
    s, cam = {"cornell": scene.cornell_box, "two_boxes": scene.cornell_two_boxes,
              "spheres4": lambda: scene.spheres_world(2.0, small_spheres=False)}[world]()
    gpu.upload_scene(s)
    for p in range(s.n_prims):
        c = s.arrays()["prim_off"][p]
        gm = s.geom[p]
        ext = 0.6 * max(gm[1] - gm[0], gm[3] - gm[2]) if s.kind[p] == capi.PRIM_RECT else 2.0 * gm[0]
        org, d = H.ray_batch(4096, 100 + p, center=c, extent=min(ext, 50.0))
        for (t0, t1) in ((0.001, FLT_MAX), (0.001, 4.0)):
            g = gpu.prim_hit_batch(p, org, d, t0, t1); o = orc.prim_hit_batch(s, p, org, d, t0, t1)
            assert (o["prim"] >= 0).sum() > 100
            H.assert_hits_equal(g, o, f"{world} prim {p}", scene=s)


def test_prim_hit_with_negative_t_min(gpu, orc):
    """The one-divide shortcut of the sphere roots decides by an estimate against a band around t_min; the band must lie around
    t_min of either sign (mirt_prim_hit_batch and the reference-order mirt_world_hit_batch take any t_min).  Rays whose first
    root falls just below a negative t_min must still find the second."""
    s, cam = scene.spheres_world(2.0, small_spheres=False)
    gpu.upload_scene(s)
    rng = np.random.default_rng(77)
    for p in (1, 2, 3):                                          # the three unit spheres
        c = s.arrays()["prim_off"][p]
        org, d = H.ray_batch(8192, 300 + p, center=c, extent=2.0)
        o0 = orc.prim_hit_batch(s, p, org, d, -1.0e30, FLT_MAX)
        first = o0["t"][o0["prim"] >= 0]
        for t_min in (-5.0, -0.75, float(np.median(first)), float(np.median(first)) * (1.0 + 4e-6)):
            g = gpu.prim_hit_batch(p, org, d, t_min, FLT_MAX); o = orc.prim_hit_batch(s, p, org, d, t_min, FLT_MAX)
            H.assert_hits_equal(g, o, f"prim {p} t_min={t_min}", scene=s)
        # t_min inside the band around each ray's own first root: both sides of it, to the last place
        hit = o0["prim"] >= 0
        for k in (-2, -1, 0, 1, 2):
            for i in np.flatnonzero(hit)[:64]:
                tm = float(np.nextafter(o0["t"][i], np.float32(np.inf if k > 0 else -np.inf)) if k else o0["t"][i])
                if abs(k) == 2:
                    tm = float(np.float32(o0["t"][i]) * np.float32(1.0 + (3e-6 if k > 0 else -3e-6)))
                g = gpu.prim_hit_batch(p, org[i:i + 1], d[i:i + 1], tm, FLT_MAX); o = orc.prim_hit_batch(s, p, org[i:i + 1], d[i:i + 1], tm, FLT_MAX)
                H.assert_hits_equal(g, o, f"prim {p} ray {i} t_min at its first root {k:+d}", scene=s)
    org, d = H.ray_batch(8192, 12, extent=6.0)
    capi.build_bvh(s); gpu.upload_scene(s)                      # one tree for both sides
    for t_min in (-3.0, -0.001):
        o, _ = orc.world_hit_batch(s, org, d, t_min, FLT_MAX, 0)
        assert (o["prim"] >= 0).sum() > 1000
        H.assert_hits_equal(gpu.world_hit_batch(org, d, t_min, FLT_MAX, 0), o, f"reference-order walk, t_min={t_min}", scene=s)


@pytest.mark.parametrize("world", ["cornell", "two_boxes", "spheres4", "spheres"])
def test_world_hit(gpu, orc, world):
    s, cam = {"cornell": scene.cornell_box, "two_boxes": scene.cornell_two_boxes,
              "spheres4": lambda: scene.spheres_world(2.0, small_spheres=False),
              "spheres": lambda: scene.spheres_world(1.5)}[world]()
    draws = np.random.default_rng(7).integers(0, 2 ** 32, 4 * s.n_objects + 8, dtype=np.uint64).astype(np.uint32)
    orc.bvh_build_reference(s, draws)                           # SimpleObjectBVHNode ctor, random axes
    gpu.upload_scene(s)                                         # the device walks this very tree
    ext = 2.0 if world != "spheres" and world != "spheres4" else 12.0
    org, d = H.ray_batch(16384, 11, extent=ext)
    for flags in (0, capi.FLAG_ACCEL_LIST):
        g = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, flags); o, st = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, flags)
        assert (o["prim"] >= 0).sum() > 1000
        H.assert_hits_equal(g, o, f"{world} flags={flags}", scene=s)
    # the library's own BVH (no tree passed): same closest hits as the list
    s.nodes = None
    gpu.upload_scene(s)
    g = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, 0); o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST)
    # coplanar surfaces (a cube standing on the floor) tie in t and the later-tested primitive wins
    # (Hitables.cpp:33,96 use <=), so the winner may differ with the visiting order; t may not
    H.assert_bits(g["t"], o["t"], "own-bvh t")
    assert np.array_equal(g["prim"] >= 0, o["prim"] >= 0)


def test_scatter(gpu, orc):
    s, cam = scene.cornell_box()
    glass2 = s.dielectric(2.4); mirror = s.metal((0.7, 0.6, 0.5), 0.0); rough = s.metal((0.9, 0.9, 0.9), 5.0)
    gpu.upload_scene(s)
    rng = np.random.default_rng(5)
    n = 4096 * len(s.mkind)
    hits = np.zeros(n, capi.HIT_DTYPE)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits["nrm"] = nrm.astype(np.float32); hits["pos"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    hits["t"] = 1.0; hits["mat"] = np.arange(n) % len(s.mkind)
    d = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(3.0)
    graze = rng.random(n) < 0.2                                  # near-tangent incidence: total internal reflection
    d[graze] = (np.cross(nrm[graze], rng.normal(size=(graze.sum(), 3))) + 0.05 * nrm[graze] * rng.normal(size=(graze.sum(), 1))).astype(np.float32)
    d[0] = 0.0                                                   # zero direction: normalize -> 0
    draws = rng.integers(0, 2 ** 32, (n, 3 * 64), dtype=np.uint64).astype(np.uint32)
    g = gpu.scatter_batch(hits, d, draws); o = orc.scatter_batch(s, hits, d, draws)
    assert np.array_equal(g["alive"], o["alive"]) and np.array_equal(g["consumed"], o["consumed"])
    for f in ("atten", "org", "dir", "emit"):
        H.assert_bits(g[f], o[f], f"scatter.{f}")
    assert o["consumed"].max() > 9 and 0 < o["alive"].mean() < 1


def test_camera_rays(gpu, orc):
    rng = np.random.default_rng(6)
    for (s_cam, aspect) in ((scene.cornell_box, 1.0), (scene.spheres_world, 1.5), (lambda a: scene.spheres_world(a, small_spheres=False), 2.0)):
        _, cam = s_cam(aspect)
        gpu.set_camera(cam)
        uv = rng.uniform(0, 1, (8192, 2)).astype(np.float32)
        draws = rng.integers(0, 2 ** 32, (8192, 128), dtype=np.uint64).astype(np.uint32)
        g = gpu.camera_rays(uv, draws); o = orc.camera_rays(cam, uv, draws)
        assert np.array_equal(g["consumed"], o["consumed"]) and o["consumed"].max() >= 6
        H.assert_bits(g["org"], o["org"], "camera.org"); H.assert_bits(g["dir"], o["dir"], "camera.dir")


WORLDS = {
    "cornell": (lambda: scene.cornell_box(1.0), 96, 96),
    "two_boxes": (lambda: scene.cornell_two_boxes(1.0), 64, 64),
    "spheres4": (lambda: scene.spheres_world(2.0, small_spheres=False), 200, 100),
    "spheres": (lambda: scene.spheres_world(1.5), 120, 80),
}


@pytest.mark.parametrize("world", list(WORLDS))
@pytest.mark.parametrize("accel", ["list", "bvh"])
def test_image_exact(gpu, orc, world, accel):
    """Same Philox keys, same arithmetic, ordered sample sum: the device image equals the oracle's
    forward-order image bit for bit (floats and RGBA8), in 1-SPP, normal-display and multi-sample mode."""
    mk, w, h = WORLDS[world]
    s, cam = mk()
    draws = np.random.default_rng(9).integers(0, 2 ** 32, 4 * s.n_objects + 8, dtype=np.uint64).astype(np.uint32)
    orc.bvh_build_reference(s, draws)
    gpu.upload_scene(s); gpu.set_camera(cam)
    af = capi.FLAG_ACCEL_LIST if accel == "list" else 0
    for (flags, spp) in ((capi.FLAG_ONE_SPP, 1), (capi.FLAG_NORMAL_DISPLAY | capi.FLAG_ONE_SPP, 1), (capi.FLAG_NORMAL_DISPLAY, 3), (0, 8)):
        g = gpu.render(w, h, spp, 50, 0x5EED, flags | af)
        osum, st = orc.trace(s, cam, w, h, spp, 50, 0x5EED, flags | af)
        opx, orgba = orc.resolve(osum, spp, flags)
        H.assert_bits(g["sum"], osum, f"{world}/{accel} flags={flags} sum")
        H.assert_bits(g["pixels"], opx, f"{world}/{accel} flags={flags} pixels")
        assert np.array_equal(g["rgba8"], orgba)
        assert g["stats"].rays == st.rays, (g["stats"].rays, st.rays)


def test_batching_and_ranges(gpu, orc):
    """Small ray-queue capacity forces pixel tiles and sample tiles; sample ranges compose."""
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    w = h = 64
    ref = gpu.render(w, h, 12, 50, 7, capi.FLAG_ACCEL_LIST)
    gpu.set_option("max_paths", 1500)                            # < w*h: pixel tiles of 1500, one sample each
    a = gpu.render(w, h, 12, 50, 7, capi.FLAG_ACCEL_LIST)
    gpu.set_option("max_paths", 3 * w * h + 17)                  # sample tiles of 3
    b = gpu.render(w, h, 12, 50, 7, capi.FLAG_ACCEL_LIST)
    gpu.set_option("max_paths", 1 << 30)
    H.assert_bits(a["sum"], ref["sum"], "pixel tiles"); H.assert_bits(b["sum"], ref["sum"], "sample tiles")
    osum, _ = orc.trace(s, cam, w, h, 12, 50, 7, capi.FLAG_ACCEL_LIST)
    H.assert_bits(ref["sum"], osum, "vs oracle")
    # [0,5) then [5,12) accumulated by the oracle in two calls == one call (ordered sum)
    part, _ = orc.trace(s, cam, w, h, 5, 50, 7, capi.FLAG_ACCEL_LIST)
    part, _ = orc.trace(s, cam, w, h, 7, 50, 7, capi.FLAG_ACCEL_LIST, spp_begin=5, sum_rgb=part)
    H.assert_bits(part, osum, "oracle ranges")


def test_launch_with_own_path_ids(gpu, orc):
    """Launches of 1 M .. 29 M paths give every wave of the persistent grid chunks of 64 path ids of its own before it
    starts claiming (mirt.hip render_persistent, TraceArgs::own_chunks).  Who traces a path changes nothing: the image of
    such a launch equals the oracle's bit for bit, for the list walk and for a BVH walk with tiled path ids, and also when
    the batches cut it into launches that take the other route (all ids claimed)."""
    for mk, w, h, spp, flags, oflags in ((lambda: scene.cornell_box(1.0), 512, 512, 6, capi.FLAG_ACCEL_LIST, capi.FLAG_ACCEL_LIST),
                                         (lambda: scene.spheres_world(480 / 320), 480, 320, 8, capi.FLAG_ACCEL_ORDERED, orc.FLAG_ACCEL_ORDERED)):
        s, cam = mk()
        if flags != capi.FLAG_ACCEL_LIST:
            capi.build_bvh(s)
        gpu.upload_scene(s); gpu.set_camera(cam)
        osum, ost = orc.trace(s, cam, w, h, spp, 50, 77, oflags)
        try:
            for max_paths in (1 << 30, 300000):                              # one launch of 1.2 - 1.6 M paths; batches of 0.3 M (no own ids)
                gpu.set_option("max_paths", max_paths)
                g = gpu.render(w, h, spp, 50, 77, flags)
                H.assert_bits(g["sum"], osum, f"{w}x{h}@{spp} max_paths={max_paths}")
                assert g["stats"].rays == ost.rays
        finally:
            gpu.set_option("max_paths", 1 << 30)


def test_tiled_path_ids(gpu, orc):
    """With a BVH walk the persistent engine enumerates pixels in 8x8 tiles (option tile_paths, rt_kernels.h tile_pixel).
    That only changes which lane traces which path: the image must equal the row-order one and the oracle's, also when
    the batches cut the image into pixel ranges that end inside a tile, and for sizes that are no multiple of 8
    (tiling off by itself)."""
    s, cam = scene.spheres_world(64 / 48)
    capi.build_bvh(s)
    gpu.upload_scene(s); gpu.set_camera(cam)
    for (w, h) in ((64, 48), (40, 24), (36, 20)):
        osum, ost = orc.trace(s, cam, w, h, 5, 50, 21, 0)
        for flags in (0, capi.FLAG_ACCEL_ORDERED):
            for tiles in (1, 0):
                gpu.set_option("tile_paths", tiles)
                for max_paths in (1 << 30, max(w * h + 13, 1024), 1029):
                    gpu.set_option("max_paths", max_paths)
                    g = gpu.render(w, h, 5, 50, 21, flags)
                    H.assert_bits(g["sum"], osum, f"{w}x{h} flags={flags} tile_paths={tiles} max_paths={max_paths}")
                    assert g["stats"].rays == ost.rays
    gpu.set_option("tile_paths", 1); gpu.set_option("max_paths", 1 << 30)


def test_yielding_walk(gpu, orc):
    """Option walk_yield: the near-first walk in resumable slices (rt_kernels.h bvh2_slice) -- a wave leaves the walk
    loop when that many lanes have finished, shades them and resumes the rest.  Only the schedule changes: images
    and ray counts equal the oracle's, and the rays per depth are the same for every setting, across batch cuts."""
    s, cam = scene.spheres_world(64 / 48)
    capi.build_bvh(s)
    gpu.upload_scene(s); gpu.set_camera(cam)
    w, h = 64, 48
    osum, ost = orc.trace(s, cam, w, h, 6, 50, 33, 0)
    ref_hist = None
    try:
        for wy in (0, 1, 7, 32, 56, 64):
            gpu.set_option("walk_yield", wy)
            for max_paths in (1 << 30, 2048):
                gpu.set_option("max_paths", max_paths)
                g = gpu.render(w, h, 6, 50, 33, capi.FLAG_ACCEL_ORDERED)
                H.assert_bits(g["sum"], osum, f"walk_yield={wy} max_paths={max_paths}")
                assert g["stats"].rays == ost.rays
                hist = list(g["stats"].depth_hist)
                ref_hist = hist if ref_hist is None else ref_hist
                assert hist == ref_hist                                      # rays per depth, whatever the schedule
        gpu.set_option("walk_yield", 3)
        g = gpu.render(w, h, 2, 2, 33, capi.FLAG_ACCEL_ORDERED)              # depth limit inside the yielding loop
        osum2, _ = orc.trace(s, cam, w, h, 2, 2, 33, 0)
        H.assert_bits(g["sum"], osum2, "walk_yield with max_depth 2")
    finally:
        gpu.set_option("walk_yield", -1); gpu.set_option("max_paths", 1 << 30)


def test_sphere_leaf_records_follow_the_scene(gpu, orc):
    """Scenes whose objects are all single unrotated spheres get 16-byte leaf records (SceneDev::sph4, rt_kernels.h
    sphere_leaf_t); any other scene keeps the generic records.  Uploaded in turn on ONE context -- spheres only, one rect among
    the spheres, spheres only again (objects that list their primitives out of order are refused) -- each renders the oracle's image
    bit for bit through the near-first walk, plain and in resumable slices."""
    w, h = 64, 48
    cam = scene.spheres_world(w / h)[1]

    def check(s, label):
        capi.build_bvh(s)
        gpu.upload_scene(s); gpu.set_camera(cam)
        osum, ost = orc.trace(s, cam, w, h, 4, 50, 5, orc.FLAG_ACCEL_ORDERED)
        for wy in (0, 56):
            gpu.set_option("walk_yield", wy)
            g = gpu.render(w, h, 4, 50, 5, capi.FLAG_ACCEL_ORDERED)
            H.assert_bits(g["sum"], osum, f"{label}, walk_yield={wy}")
            assert g["stats"].rays == ost.rays

    try:
        s1 = scene.spheres_world(w / h)[0]
        check(s1, "spheres only")
        s2 = scene.spheres_world(w / h)[0]
        s2.rect(scene.XZ_RECT, (0.0, 0.6, 0.0), (0.0, 0.0, 0.0), 3.0, 3.0, 0, s2.lambertian((0.7, 0.2, 0.2)))
        check(s2, "one rect among the spheres")
        s3 = scene.spheres_world(w / h)[0]                                 # objects out of primitive order are refused at upload, so
        s3.obj_first[1], s3.obj_first[2] = s3.obj_first[2], s3.obj_first[1]; s3._arrays = None   # "object index == primitive index" holds
        with pytest.raises(capi.MirtError):                                # wherever the sphere records are used
            gpu.upload_scene(s3)
        check(scene.spheres_world(w / h)[0], "spheres only, after the other two")
        # the sliced walk with the top of the tree in LDS (k_trace_bvh<4, ...>, option lds_top; by default only trees of 65536 nodes
        # and more): the same image whatever part of the tree the LDS copy holds, down to "all of it"
        for top in (1, 7, 32, 256):
            gpu.set_option("lds_top", top)
            check(scene.spheres_world(w / h)[0], f"lds_top={top}")
            check(s2, f"one rect among the spheres, lds_top={top}")
        with pytest.raises(capi.MirtError):
            gpu.set_option("lds_top", 257)
    finally:
        gpu.set_option("walk_yield", -1); gpu.set_option("lds_top", -1)


def test_large_tree_variants_agree(gpu, orc):
    """A tree of more than 65536 nodes (90 000 spheres) switches the sliced walk to its variant with the top 32 nodes in LDS
    (k_trace_bvh<4, ...>).  The persistent engine as it chooses, sliced, plain, and the wavefront engine render the bits of the
    oracle walking the same tree near-first, and count the same rays."""
    w, h = 256, 144
    s, cam = scene.spheres_world(w / h, grid=150, look_from=(26.0, 6.0, 6.0))
    assert s.n_prims > 65537
    capi.build_bvh(s)
    gpu.upload_scene(s); gpu.set_camera(cam)
    osum, ost = orc.trace(s, cam, w, h, 8, 50, 7, orc.FLAG_ACCEL_ORDERED)
    try:
        ref = (osum, ost.rays)
        for label, opts in (("default", {"engine": 1, "walk_yield": -1}), ("sliced", {"walk_yield": 56}), ("plain", {"walk_yield": 0}),
                            ("wavefront", {"engine": 0})):
            for k, v in opts.items():
                gpu.set_option(k, v)
            g = gpu.render(w, h, 8, 50, 7, capi.FLAG_ACCEL_AUTO, want=("sum",))
            cur = (np.asarray(g["sum"]).copy(), g["stats"].rays)
            H.assert_bits(cur[0], ref[0], f"90 000 spheres, {label}")
            assert cur[1] == ref[1]
    finally:
        gpu.set_option("engine", 1); gpu.set_option("walk_yield", -1)


def test_leaf_wait_changes_no_bit(gpu, orc):
    """Option leaf_wait (rt_kernels.h closest_bvh2 / bvh2_slice): how many lanes of a wave must stand on an object before
    the wave tests objects.  A waiting lane does nothing, so every ray sees the same boxes, objects and t_max in the same
    order whatever the setting: hit records incl. tie winners equal the ordered oracle's, images equal bit for bit, plain and
    yielding walk alike."""
    s, cam = scene.spheres_world(64 / 48)
    capi.build_bvh(s)
    gpu.upload_scene(s); gpu.set_camera(cam)
    w, h = 64, 48
    org, d = H.ray_batch(4096, 77, center=(0.0, 1.0, 0.0), extent=12.0)
    o_hits, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, orc.FLAG_ACCEL_ORDERED)
    osum, ost = orc.trace(s, cam, w, h, 5, 50, 9, orc.FLAG_ACCEL_ORDERED)
    try:
        for lw in (1, 2, 4, 17, 64):
            gpu.set_option("leaf_wait", lw)
            for wy in (0, 40):
                gpu.set_option("walk_yield", wy)
                g = gpu.render(w, h, 5, 50, 9, capi.FLAG_ACCEL_ORDERED)
                H.assert_bits(g["sum"], osum, f"leaf_wait={lw} walk_yield={wy}")
                assert g["stats"].rays == ost.rays
        H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_ORDERED), o_hits, "hit batch, ordered walk", scene=s)
        with pytest.raises(capi.MirtError):
            gpu.set_option("leaf_wait", 0)
    finally:
        gpu.set_option("leaf_wait", -1); gpu.set_option("walk_yield", -1)


def test_depth_limits(gpu, orc):
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    for md in (0, 1, 3):
        g = gpu.render(48, 48, 4, md, 3, capi.FLAG_ACCEL_LIST)
        osum, st = orc.trace(s, cam, 48, 48, 4, md, 3, capi.FLAG_ACCEL_LIST)
        H.assert_bits(g["sum"], osum, f"max_depth={md}")


def test_errors(gpu):
    s, cam = scene.cornell_box(1.0)
    bad = scene.FlatScene()
    with pytest.raises(capi.MirtError):
        gpu.upload_scene(bad)                                     # empty scene
    s.pmat[0] = 99
    s._arrays = None
    with pytest.raises(capi.MirtError, match="material index"):
        gpu.upload_scene(s)
    # ABI guard: structs that claim another size are refused with an error code, and a short mirt_stats is never written
    import ctypes as C
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    p = capi.RenderParams(8, 8, 0, 1, 5, 0, 1)
    st = capi.Stats(); st.struct_size -= 512; st.rays = 77
    out = np.zeros((8, 8, 4), np.uint8)
    assert gpu.lib.mirt_render(gpu.h, C.byref(p), None, None, out.ctypes.data_as(C.c_void_p), C.byref(st)) == -1
    assert b"mirt_stats.struct_size" in gpu.lib.mirt_last_error(gpu.h) and st.rays == 77 and not out.any()
    p.struct_size += 4
    assert gpu.lib.mirt_render(gpu.h, C.byref(p), None, None, out.ctypes.data_as(C.c_void_p), None) == -1 and not out.any()
    bad = s.as_struct(); bad.struct_size = 0
    assert gpu.lib.mirt_upload_scene(gpu.h, C.byref(bad)) == -1


@pytest.mark.parametrize("world", ["cornell", "spheres4", "spheres", "two_boxes"])
def test_ordered_traversal_and_own_bvh(gpu, orc, world):
    """The library's own SAH tree walked near-first (what ACCEL_AUTO uses for larger scenes): same closest t as
    the list for every ray; same image as the list where no two surfaces are coplanar (two_boxes has a cube
    standing on the floor, where the tie winner legitimately depends on the visiting order)."""
    mk, w, h = WORLDS[world]
    s, cam = mk()
    s.nodes = None                                               # no tree passed: mirt_upload_scene builds its own
    gpu.upload_scene(s); gpu.set_camera(cam)
    org, d = H.ray_batch(16384, 21, extent=2.0 if world in ("cornell", "two_boxes") else 12.0)
    g = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
    o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST)
    H.assert_bits(g["t"], o["t"], "ordered t")
    assert np.array_equal(g["prim"] >= 0, o["prim"] >= 0)
    if world != "two_boxes":
        assert np.array_equal(g["prim"], o["prim"])
        a = gpu.render(w, h, 6, 50, 0x5EED, capi.FLAG_ACCEL_ORDERED)
        osum, st = orc.trace(s, cam, w, h, 6, 50, 0x5EED, capi.FLAG_ACCEL_LIST)
        H.assert_bits(a["sum"], osum, "ordered image")
        assert a["stats"].rays == st.rays
    # the oracle's restatement of the near-first walk on the library's own tree: whole records, tie winners included
    capi.build_bvh(s)
    o2, st2 = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, orc.FLAG_ACCEL_ORDERED)
    H.assert_hits_equal(g, o2, "ordered walk vs its restatement", scene=s)
    a = gpu.render(w, h, 3, 50, 7, capi.FLAG_ACCEL_ORDERED)
    osum, st3 = orc.trace(s, cam, w, h, 3, 50, 7, orc.FLAG_ACCEL_ORDERED)
    H.assert_bits(a["sum"], osum, "ordered image vs ordered oracle")
    assert a["stats"].rays == st3.rays and 0 < st3.nodes_visited
    # the same tree, host-built through the C ABI, walked in the reference's order by oracle and device alike
    depth = capi.build_bvh(s)
    assert 1 <= depth <= 40 and len(s.nodes[2]) == max(1, s.n_objects - 1)
    gpu.upload_scene(s)
    g1 = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, 0)
    o1, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, 0)
    H.assert_hits_equal(g1, o1, "own tree, reference order", scene=s)


@pytest.mark.parametrize("world", ["cornell", "spheres4", "spheres"])
def test_axis_parallel_rays(gpu, orc, world):
    """Rays with zero direction components (1/d = inf, 0 * inf = NaN inside the slab tests) from origins on lattice
    points, many of them on the planes of rects and of node boxes: the list walk, the reference-order walk and the
    near-first walk (fmax/fmin slab test, one final compare) must all agree with the oracle."""
    mk, w, h = WORLDS[world]
    s, cam = mk()
    s.nodes = None
    capi.build_bvh(s)                                              # the library's own tree, for oracle and device alike
    gpu.upload_scene(s)
    rng = np.random.default_rng(77)
    n = 16384
    lat = 0.5 if world == "cornell" else 1.0
    org = (rng.integers(-6, 7, (n, 3)) * lat).astype(np.float32)
    d = rng.integers(-1, 2, (n, 3)).astype(np.float32)
    d[(d == 0).all(axis=1)] = (0.0, -1.0, 0.0)
    mix = rng.random(n) < 0.3                                      # some with one free component
    d[mix, rng.integers(0, 3, mix.sum())] = rng.standard_normal(mix.sum()).astype(np.float32)
    d[::7] = np.where(d[::7] == 0, np.float32(-0.0), d[::7])       # both signs of zero
    o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST)
    assert (o["prim"] >= 0).sum() > 250
    H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST), o, "axis-parallel list", scene=s)
    # the reference's BVH walk culls a ray that only grazes a node box (AABB.cpp:17 uses <=), so it misses some
    # edge and corner hits the list finds; the device does the same in the same cases
    o1, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, 0)
    H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, 0), o1, "axis-parallel reference order", scene=s)
    # near-first: the whole hit record equals the oracle's walk of the same tree in the same order.  Against the
    # reference-order walk the closest t is the same except on grazing rays: the reference's walk, having met the floor
    # first, culls the box of a ball that touches the floor there (AABB.cpp:17, `<=`) and so misses the ball's hit an ulp
    # nearer; the near-first walk meets the ball first and agrees with the list.  A few rays of this lattice in 16 384.
    g = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
    o2, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
    H.assert_hits_equal(g, o2, "axis-parallel near-first", scene=s)
    assert np.array_equal(g["prim"] >= 0, o1["prim"] >= 0)
    differs = g["t"].view(np.uint32) != o1["t"].view(np.uint32)
    assert differs.sum() <= 8 and np.array_equal(g["t"].view(np.uint32)[differs], o["t"].view(np.uint32)[differs]) and (g["t"][differs] < o1["t"][differs]).all()


def test_sphere_leaf_quotients_at_every_scale(gpu, orc):
    """The near-first walks take a sphere leaf's root through div_by when the wave's operands lie in its range (rt_device.h
    prim_t: by2a; 2a in [2^-20, 2^20], first product below 2^19) and through the plain division otherwise, whole wave.  Rays
    whose directions are scaled by 2^-40 .. 2^40 -- waves of one scale and waves that mix them -- and origins far outside the
    scene hit what the oracle hits, bit for bit, at the render's t_min and at a t_min below div_by's promise."""
    s, cam = scene.spheres_world(1.5)
    s.nodes = None
    capi.build_bvh(s)
    gpu.upload_scene(s)
    rng = np.random.default_rng(4242)
    n = 32768
    org = (rng.standard_normal((n, 3)) * 4.0).astype(np.float32)
    org[:, 1] = np.abs(org[:, 1]) + 0.3
    tgt = (rng.standard_normal((n, 3)) * 3.0).astype(np.float32); tgt[:, 1] = 0.3
    d = tgt - org
    scales = np.array([2.0 ** e for e in (-40, -24, -11, -9, -3, 0, 3, 9, 11, 24, 40)], dtype=np.float32)
    k = np.empty(n, dtype=np.int64)
    k[: n // 2] = np.repeat(np.arange(n // 2 // 64) % len(scales), 64)         # whole waves of one scale
    k[n // 2:] = rng.integers(0, len(scales), n - n // 2)                      # mixed waves
    d = (d * scales[k][:, None]).astype(np.float32)
    far = rng.random(n) < 0.05                                                 # numerators beyond 2^40
    org[far] = (org[far] * np.float32(2.0 ** 30)).astype(np.float32)
    for t_min in (0.001, 1e-7):
        o, _ = orc.world_hit_batch(s, org, d, t_min, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
        assert (o["prim"] >= 0).sum() > n // 4
        g = gpu.world_hit_batch(org, d, t_min, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
        H.assert_hits_equal(g, o, f"sphere leaves, directions at every scale, t_min={t_min}", scene=s)


def test_facing_rect_pairs(gpu, orc):
    """closest_list tests the lowest and the highest plane of an orientation class as ONE rect when every ray of a wave starts
    between them (rt_kernels.h rect_pair).  A room with a pair in each of the three orientations, more rects between the pairs,
    a partition that does not span the room and two balls; ray origins inside the room, exactly on every plane, one float to
    either side of every plane, and outside the room heading in -- in runs of 64 of a kind, so that whole waves take the merged
    test, and shuffled, so that waves mix them and fall back.  Hit records against the oracle's list walk, and an image."""
    s = scene.FlatScene(ambient=(0.2, 0.3, 0.4))
    grey, red, lamp, glass = s.lambertian((0.6, 0.6, 0.6)), s.lambertian((0.7, 0.2, 0.2)), s.diffuse_light((6.0, 6.0, 6.0)), s.dielectric(1.5)
    z = (0.0, 0.0, 0.0)
    walls = [(scene.ZY_RECT, (2.0, 0.0, 0.0), True), (scene.ZY_RECT, (-1.5, 0.0, 0.0), False),       # x pair (not symmetric)
             (scene.XZ_RECT, (0.0, 1.25, 0.0), True), (scene.XZ_RECT, (0.0, -1.0, 0.0), False),      # y pair
             (scene.XY_RECT, (0.0, 0.0, -2.0), False), (scene.XY_RECT, (0.0, 0.0, 1.75), True)]      # z pair
    for axes, c, rev in walls:
        s.rect(axes, c, z, 6.0, 6.0, rev, grey)
    s.rect(scene.XZ_RECT, (0.0, 1.2, 0.0), z, 1.0, 1.0, True, lamp)                                   # a third plane in the y class
    s.rect(scene.ZY_RECT, (0.5, -0.25, 0.0), z, 1.0, 1.5, False, red)                                 # a partition inside the x pair
    s.rect(scene.XY_RECT, (0.25, 0.0, 0.0), z, 1.5, 1.0, False, red)
    s.sphere((-0.7, -0.6, 0.3), 0.4, glass); s.sphere((1.2, -0.7, -0.8), 0.3, grey)
    capi.build_bvh(s)
    gpu.upload_scene(s)
    rng = np.random.default_rng(123)
    n = 64 * 96
    planes = {0: (-1.5, 2.0, 0.5), 1: (-1.0, 1.25, 1.2), 2: (-2.0, 1.75, 0.0)}
    org = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32) * np.float32([1.4, 0.9, 1.6])
    d = rng.standard_normal((n, 3)).astype(np.float32)
    kind = (np.arange(n) // 64) % 6                                # runs of 64: inside, on a plane, an ulp outside, an ulp inside, outside heading in, outside
    for i in range(n):
        ax = int(rng.integers(0, 3)); pl = np.float32(planes[ax][int(rng.integers(0, 3))])
        if kind[i] == 1: org[i, ax] = pl
        elif kind[i] == 2: org[i, ax] = np.nextafter(pl, np.float32(np.inf) if pl > 0 else np.float32(-np.inf))
        elif kind[i] == 3: org[i, ax] = np.nextafter(pl, np.float32(0.0))
        elif kind[i] >= 4:
            org[i, ax] = np.float32(pl * rng.uniform(1.5, 4.0)) if pl != 0 else np.float32(5.0)
            if kind[i] == 4: d[i, ax] = -np.sign(org[i, ax]) * abs(d[i, ax])
    for tag, order in (("runs of a kind", np.arange(n)), ("shuffled", rng.permutation(n))):
        o, _ = orc.world_hit_batch(s, org[order], d[order], 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST)
        assert (o["prim"] >= 0).mean() > 0.7 and len(set(o["prim"].tolist())) >= 11
        H.assert_hits_equal(gpu.world_hit_batch(org[order], d[order], 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST), o, f"facing pairs, {tag}", scene=s)
    # a t_min of either size: 2^-19 is the least the shared-division path (and with it the merged test) accepts
    for t_min in (2.0 ** -19, 2.0 ** -21, 0.25):
        o, _ = orc.world_hit_batch(s, org, d, t_min, 50.0, capi.FLAG_ACCEL_LIST)
        H.assert_hits_equal(gpu.world_hit_batch(org, d, t_min, 50.0, capi.FLAG_ACCEL_LIST), o, f"facing pairs, t_min {t_min}", scene=s)
    cam = scene.make_camera((1.2, 0.4, 1.5), (-0.4, -0.3, -0.6), 70.0, 0.05, 1.5)
    gpu.set_camera(cam)
    osum, ost = orc.trace(s, cam, 96, 64, 8, 50, 77, orc.FLAG_ACCEL_LIST)
    g = gpu.render(96, 64, 8, 50, 77, capi.FLAG_ACCEL_LIST)
    H.assert_bits(g["sum"], osum, "room image"); assert g["stats"].rays == ost.rays and ost.rays > 4 * 96 * 64 * 8


def test_million_spheres(gpu, orc):
    """BASELINE config 4 (the generator of World.cpp:36-58 extended to a 1000x1000 grid, ~1M spheres): the host-built
    SAH tree is walked in the reference's order and near-first by the oracle and by the device (bit-exact hits and images,
    each order against its own twin; the two orders agree on the closest t of the ray batch and on all but isolated pixels)."""
    s, cam = scene.spheres_world(1.5, grid=500, look_from=(26.0, 6.0, 6.0))
    assert s.n_objects > 900000
    depth = capi.build_bvh(s)
    assert 20 <= depth <= 60
    gpu.upload_scene(s); gpu.set_camera(cam)
    org, d = H.ray_batch(8192, 31, extent=40.0)
    d[:, 1] = -np.abs(d[:, 1])                                  # look down at the field
    o, st = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, 0)
    assert (o["prim"] > 3).sum() > 200                           # small spheres are being hit, not just the ground
    H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, 0), o, "1M ref order", scene=s)
    g2 = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
    H.assert_bits(g2["t"], o["t"], "1M near-first t"); assert np.array_equal(g2["prim"], o["prim"])
    w, h = 96, 64
    sums = []
    for flags in (0, capi.FLAG_ACCEL_ORDERED):                   # each walk against the oracle's walk in the same order
        osum, ost = orc.trace(s, cam, w, h, 2, 50, 11, flags)
        g = gpu.render(w, h, 2, 50, 11, flags)
        H.assert_bits(g["sum"], osum, f"1M image flags={flags}")
        assert g["stats"].rays == ost.rays
        sums.append(g["sum"])
    # the two orders see the same surfaces; they may disagree only on the reference sphere test's false positives (a ray that
    # passes just outside a sphere is accepted when its box happens to be visited): isolated pixels
    assert (np.abs(sums[0] - sums[1]).max(axis=2) > 0).sum() <= 4


def test_material_stream_build_is_bit_identical(tmp_path):
    """lib/libmirt_defer.so (build flag MIRT_DEFER_MD: metal and glass vertices parked in a per-wave queue and shaded 64 at a
    time, profiles/r02_material_stream.txt) regroups work, never arithmetic: sums, ray counts and depth histograms equal the
    product's bit for bit.  Each build renders in its own process (MIRT_LIB selects the library)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(os.path.dirname(capi.LIB_PATH), "libmirt_defer.so")
    if not os.path.exists(variant):
        pytest.skip("libmirt_defer.so was not built (MIRT_SKIP_VARIANT)")
    for world, w, h, spp in (("cornell", 192, 192, 24), ("spheres4", 200, 100, 40), ("two_boxes", 128, 128, 8)):
        got = []
        for lib in (capi.LIB_PATH, variant):
            out = str(tmp_path / f"{world}_{os.path.basename(lib)}.npz")
            subprocess.run([sys.executable, os.path.join(root, "tools", "ab_image.py"), out, world, str(w), str(h), str(spp)],
                           env=dict(os.environ, MIRT_LIB=lib), check=True, timeout=120, stdout=subprocess.DEVNULL)
            got.append(np.load(out))
        a, b = got
        H.assert_bits(a["sum"], b["sum"], f"{world}: second-stream build")
        assert int(a["rays"]) == int(b["rays"]) and np.array_equal(a["hist"], b["hist"])


def test_two_paths_per_lane_is_bit_identical(gpu, orc):
    """Option paths_per_lane = 2 (k_trace2: two paths in every lane share the metal/dielectric pass, the cooperative rounds and
    the record fetches of the list walk; measured slower, profiles/r02_two_paths.txt, hence not the default) regroups work,
    never arithmetic: sums, ray counts and depth histograms equal the oracle's and the default kernel's bit for bit, also
    with path counts that leave path slots empty, with three sample tiles, and at depth limits 0 and 2."""
    for world, (w, h, spp, md) in (("cornell", (96, 96, 9, 50)), ("two_boxes", (64, 64, 5, 50)), ("spheres4", (200, 100, 6, 50)),
                                   ("cornell", (7, 3, 1, 50)), ("cornell", (65, 1, 2, 0)), ("cornell", (33, 31, 7, 2))):
        s, cam = WORLDS[world][0]()
        s.nodes = None
        gpu.upload_scene(s); gpu.set_camera(cam)
        out = {}
        try:
            for ppl in (1, 2):
                gpu.set_option("paths_per_lane", ppl)
                out[ppl] = gpu.render(w, h, spp, md, 91, capi.FLAG_ACCEL_LIST)
            gpu.set_option("max_paths", max(1024, 2 * w * h + 3))
            out["tiled"] = gpu.render(w, h, spp, md, 91, capi.FLAG_ACCEL_LIST)
        finally:
            gpu.set_option("paths_per_lane", 1); gpu.set_option("max_paths", 1 << 30)
        osum, ost = orc.trace(s, cam, w, h, spp, md, 91, capi.FLAG_ACCEL_LIST)
        for key in (2, "tiled"):
            H.assert_bits(out[key]["sum"], osum, f"{world} {w}x{h}@{spp} paths_per_lane 2 ({key}) vs oracle")
            H.assert_bits(out[key]["sum"], out[1]["sum"], f"{world} {w}x{h}@{spp} paths_per_lane 2 ({key}) vs 1")
            assert out[key]["stats"].rays == ost.rays == out[1]["stats"].rays
            assert list(out[key]["stats"].depth_hist) == list(out[1]["stats"].depth_hist)
    # metal and glass only: every vertex wants the shared pass, so in most lanes both paths do and one of them must wait
    s = scene.FlatScene(ambient=(0.7, 0.8, 1.0))
    mats = [s.metal((0.8, 0.6, 0.4), 0.3), s.dielectric(1.5), s.metal((0.9, 0.9, 0.9), 0.0)]
    for i, c in enumerate(((0.0, -100.5, -1.0), (0.0, 0.0, -1.0), (1.0, 0.0, -1.0), (-1.0, 0.0, -1.0), (0.3, 0.4, 0.2))):
        s.sphere(c, 100.0 if i == 0 else (0.5 if i < 4 else 0.2), mats[i % 3])
    cam = scene.make_camera((0.0, 0.6, 2.5), (0.0, 0.0, -1.0), 40.0, 0.0, 1.5)
    gpu.upload_scene(s); gpu.set_camera(cam)
    try:
        gpu.set_option("paths_per_lane", 2)
        g = gpu.render(90, 60, 7, 50, 5, capi.FLAG_ACCEL_LIST)
    finally:
        gpu.set_option("paths_per_lane", 1)
    osum, ost = orc.trace(s, cam, 90, 60, 7, 50, 5, capi.FLAG_ACCEL_LIST)
    H.assert_bits(g["sum"], osum, "metal and glass only, paths_per_lane 2")
    assert g["stats"].rays == ost.rays


@pytest.mark.parametrize("world", list(WORLDS))
def test_engines_identical(gpu, orc, world):
    """The persistent engine (k_trace, default) and the wavefront engine (k_bounce per depth) run the same per-path
    arithmetic with the same Philox keys: same bits, same ray counts per depth, under any batching."""
    mk, w, h = WORLDS[world]
    s, cam = mk()
    s.nodes = None
    gpu.upload_scene(s); gpu.set_camera(cam)
    out = {}
    try:
        for engine in (1, 0):
            gpu.set_option("engine", engine)
            for flags in (capi.FLAG_ACCEL_AUTO, capi.FLAG_ACCEL_LIST, 0):
                out[engine, flags] = gpu.render(w, h, 5, 50, 77, flags)
            gpu.set_option("max_paths", 2 * w * h + 3)                   # three sample tiles
            out[engine, "tiled"] = gpu.render(w, h, 5, 50, 77, capi.FLAG_ACCEL_LIST)
            gpu.set_option("max_paths", 1 << 30)
    finally:
        gpu.set_option("engine", 1); gpu.set_option("max_paths", 1 << 30)
    osum, ost = orc.trace(s, cam, w, h, 5, 50, 77, capi.FLAG_ACCEL_LIST)
    for key, g in out.items():
        if world != "two_boxes" or key[1] in (capi.FLAG_ACCEL_LIST, "tiled"):   # coplanar ties: see test_ordered_traversal_and_own_bvh
            H.assert_bits(g["sum"], osum, f"{world} {key}")
            assert g["stats"].rays == ost.rays
    for flags in (capi.FLAG_ACCEL_AUTO, capi.FLAG_ACCEL_LIST, 0, "tiled"):
        a, b = out[1, flags], out[0, flags]
        H.assert_bits(a["sum"], b["sum"], f"{world} engines {flags}")
        assert list(a["stats"].depth_hist) == list(b["stats"].depth_hist) and a["stats"].rays == b["stats"].rays


def test_persistent_engine_edges(gpu, orc):
    """Path counts that are not multiples of the wave, of the claim size or of the stash; max_depth 0; one pixel."""
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    for (w, h, spp, md) in ((1, 1, 1, 50), (7, 3, 1, 50), (13, 5, 3, 2), (65, 1, 2, 0), (33, 31, 7, 50)):
        g = gpu.render(w, h, spp, md, 5, capi.FLAG_ACCEL_LIST)
        osum, ost = orc.trace(s, cam, w, h, spp, md, 5, capi.FLAG_ACCEL_LIST)
        H.assert_bits(g["sum"], osum, f"{w}x{h}@{spp} depth {md}")
        assert g["stats"].rays == ost.rays and g["stats"].samples == w * h * spp
        hist = list(g["stats"].depth_hist)
        assert hist[0] == w * h * spp and sum(hist) == ost.rays and all(a >= b for a, b in zip(hist, hist[1:])) and not any(hist[md + 1:])


def test_launch_regimes_by_size(gpu, orc):
    """render_persistent picks grid size, claim size and the statically dealt share of the path ids by launch size
    (profiles/r02_small_launches.txt).  Every path id of a batch must be traced exactly once in each regime: sums, ray counts
    and path-length histograms against the oracle for the list walk (capped grid with claims of 64; claims of 256 with four and
    with three eighths static; path counts that are no multiple of anything) and for the BVH walk (capped grid; claims of 64
    and of 128 with three eighths static)."""
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    for (w, h, spp) in ((640, 480, 3), (1021, 509, 5), (1024, 1024, 13)):          # 0.9 M, 2.6 M, 13.6 M paths
        g = gpu.render(w, h, spp, 50, 9, capi.FLAG_ACCEL_LIST, want=("sum",))
        osum, ost = orc.trace(s, cam, w, h, spp, 50, 9, capi.FLAG_ACCEL_LIST)
        H.assert_bits(g["sum"], osum, f"list walk {w}x{h}@{spp}")
        assert g["stats"].rays == ost.rays and list(g["stats"].depth_hist)[0] == w * h * spp
    s, cam = scene.spheres_world(1.5)
    capi.build_bvh(s)
    gpu.upload_scene(s); gpu.set_camera(cam)
    for (w, h, spp) in ((600, 400, 2), (600, 400, 11), (1203, 797, 8)):              # 0.5 M, 2.6 M, 7.7 M paths
        g = gpu.render(w, h, spp, 50, 9, capi.FLAG_ACCEL_ORDERED, want=("sum",))
        osum, ost = orc.trace(s, cam, w, h, spp, 50, 9, capi.FLAG_ACCEL_ORDERED)
        H.assert_bits(g["sum"], osum, f"near-first walk {w}x{h}@{spp}")
        assert g["stats"].rays == ost.rays and list(g["stats"].depth_hist)[0] == w * h * spp


def random_scene(seed):
    """Spheres, rotated rects of every alignment and rotated cubes with all four material kinds, at random places."""
    rng = np.random.default_rng(seed)
    s = scene.FlatScene(ambient=tuple(rng.uniform(0.2, 1.0, 3)))
    mats = [s.lambertian(tuple(rng.uniform(0.1, 0.9, 3))) for _ in range(3)] + [s.metal(tuple(rng.uniform(0.3, 0.9, 3)), float(rng.uniform(0, 0.6))),
            s.metal((0.8, 0.8, 0.8), 0.0), s.dielectric(float(rng.uniform(1.2, 2.2))), s.diffuse_light(tuple(rng.uniform(1, 6, 3)))]
    for _ in range(int(rng.integers(3, 9))):
        s.sphere(tuple(rng.uniform(-2, 2, 3)), float(rng.uniform(0.2, 0.9)), mats[int(rng.integers(len(mats)))])
    for _ in range(int(rng.integers(3, 8))):
        rot = tuple(rng.uniform(-1.2, 1.2, 3)) if rng.random() < 0.7 else (0.0, 0.0, 0.0)
        s.rect(int(rng.integers(3)), tuple(rng.uniform(-2.5, 2.5, 3)), rot, float(rng.uniform(0.5, 4)), float(rng.uniform(0.5, 4)), bool(rng.integers(2)), mats[int(rng.integers(len(mats)))])
    for _ in range(int(rng.integers(0, 3))):
        s.cube(tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(-1, 1, 3)), tuple(rng.uniform(0.3, 1.5, 3)), mats[int(rng.integers(len(mats)))])
    eye = rng.uniform(-1, 1, 3); eye = eye / np.linalg.norm(eye) * rng.uniform(5, 8)
    cam = scene.make_camera(tuple(eye), tuple(rng.uniform(-0.5, 0.5, 3)), float(rng.uniform(25, 50)), float(rng.uniform(0, 0.3)) if rng.random() < 0.5 else 0.0, 1.5)
    return s, cam


@pytest.mark.parametrize("seed", range(int(os.environ.get("MIRT_RANDOM_SEEDS", "8"))))      # a longer campaign: MIRT_RANDOM_SEEDS=96
def test_random_scenes(gpu, orc, seed):
    """Beyond the shipped worlds: random mixtures of every primitive, rotation and material.  Reference-order walk of the
    reference-built tree and the list are bit-exact against the oracle (hits and images); the near-first walk of the
    library's own tree finds the same closest t."""
    s, cam = random_scene(100 + seed)
    draws = np.random.default_rng(seed).integers(0, 2 ** 32, 4 * s.n_objects + 8, dtype=np.uint64).astype(np.uint32)
    orc.bvh_build_reference(s, draws)
    gpu.upload_scene(s); gpu.set_camera(cam)
    org, d = H.ray_batch(8192, 40 + seed, extent=5.0)
    for flags in (0, capi.FLAG_ACCEL_LIST):
        o, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, flags)
        assert (o["prim"] >= 0).sum() > 250
        H.assert_hits_equal(gpu.world_hit_batch(org, d, 0.001, FLT_MAX, flags), o, f"seed {seed} flags {flags}", scene=s)
    w, h = 72, 48
    for flags in (0, capi.FLAG_ACCEL_LIST):
        g = gpu.render(w, h, 4, 50, seed, flags)
        osum, ost = orc.trace(s, cam, w, h, 4, 50, seed, flags)
        H.assert_bits(g["sum"], osum, f"seed {seed} image flags {flags}")
        assert g["stats"].rays == ost.rays
        if flags == capi.FLAG_ACCEL_LIST:                                  # the same list walk with two paths in every lane (k_trace2)
            try:
                gpu.set_option("paths_per_lane", 2)
                g = gpu.render(w, h, 4, 50, seed, flags)
            finally:
                gpu.set_option("paths_per_lane", 1)
            H.assert_bits(g["sum"], osum, f"seed {seed} image, two paths per lane")
            assert g["stats"].rays == ost.rays
    olist, _ = orc.world_hit_batch(s, org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_LIST)
    s.nodes = None
    gpu.upload_scene(s)
    g2 = gpu.world_hit_batch(org, d, 0.001, FLT_MAX, capi.FLAG_ACCEL_ORDERED)
    H.assert_bits(g2["t"], olist["t"], f"seed {seed} near-first t")
    # the library's own tree walked near-first, plain and in resumable slices (the default of large launches), against the
    # oracle walking that tree in that order
    capi.build_bvh(s)
    gpu.upload_scene(s)
    osum, ost = orc.trace(s, cam, w, h, 4, 50, seed, orc.FLAG_ACCEL_ORDERED)
    try:
        for wy in (0, 56):
            gpu.set_option("walk_yield", wy)
            g = gpu.render(w, h, 4, 50, seed, capi.FLAG_ACCEL_ORDERED)
            H.assert_bits(g["sum"], osum, f"seed {seed} image, own tree near-first, walk_yield={wy}")
            assert g["stats"].rays == ost.rays
    finally:
        gpu.set_option("walk_yield", -1)
