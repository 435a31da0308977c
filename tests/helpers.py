"""Shared helpers of the parity tests: seeded ray batches and comparisons."""
import numpy as np


def ray_batch(n, seed, center=(0, 0, 0), extent=2.0):
    """Seeded mix: rays aimed at the region of interest, random rays, axis-parallel directions with
    exact zero components (also -0), origins inside the region, unnormalised and tiny directions."""
    rng = np.random.default_rng(seed)
    c = np.asarray(center, np.float32)
    org = (c + rng.uniform(-3 * extent, 3 * extent, (n, 3))).astype(np.float32)
    tgt = (c + rng.uniform(-extent, extent, (n, 3))).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    k = n // 8
    d[:k] = rng.normal(size=(k, 3)).astype(np.float32)                       # arbitrary directions
    axis = rng.integers(0, 3, k)
    d[k:2 * k] = 0.0
    d[np.arange(k, 2 * k), axis] = rng.choice([-1.0, 1.0, 0.5, -2.0], k)     # axis-parallel, zero components
    neg0 = rng.random(k) < 0.5
    d[k:2 * k][neg0, (axis[neg0] + 1) % 3] = -0.0
    org[2 * k:3 * k] = (c + rng.uniform(-0.3 * extent, 0.3 * extent, (k, 3))).astype(np.float32)   # origin inside
    d[3 * k:4 * k] *= np.float32(1e-3)                                       # short directions
    d[4 * k:5 * k] *= np.float32(37.5)                                       # long directions
    return np.ascontiguousarray(org), np.ascontiguousarray(d)


def canon(a):
    """Bit pattern with -0 folded to +0 and all NaNs to one pattern (identity rotations may differ in
    the sign of a zero, see rt_device.h to_leaf_space)."""
    a = np.ascontiguousarray(a, np.float32).copy()
    a[a == 0] = 0.0
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def assert_bits(a, b, what=""):
    ca, cb = canon(a), canon(b)
    if not np.array_equal(ca, cb):
        bad = np.argwhere(ca != cb)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {bad.shape[0]} of {ca.size} values differ; first at {i}: "
                             f"{np.asarray(a, np.float32)[i]!r} vs {np.asarray(b, np.float32)[i]!r}")


def assert_hits_equal(g, o, what="", uv_tol=None):
    """Hit records: everything bit-exact; sphere u,v (atan2f/asinf, libm-dependent) within uv_tol."""
    assert np.array_equal(g["prim"], o["prim"]), f"{what}: winning primitive differs in {(g['prim'] != o['prim']).sum()} rays"
    assert np.array_equal(g["mat"], o["mat"]), f"{what}: material differs"
    for f in ("t", "pos", "nrm"):
        assert_bits(g[f], o[f], f"{what}.{f}")
    for f in ("u", "v"):
        if uv_tol is None:
            assert_bits(g[f], o[f], f"{what}.{f}")
        else:
            assert np.allclose(g[f], o[f], rtol=0, atol=uv_tol), f"{what}.{f}: max diff {np.abs(g[f] - o[f]).max()}"


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
