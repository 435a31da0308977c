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


def canon(a, fold_zero=False):
    """Bit patterns with all NaNs folded to one pattern.  Strict about the sign of a zero unless fold_zero (see
    assert_hits_equal: the one place where the device may legitimately differ in it)."""
    a = np.ascontiguousarray(a, np.float32).copy()
    if fold_zero:
        a[a == 0] = 0.0
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def assert_bits(a, b, what="", fold_zero=False):
    """Bit-exact, the sign of a zero included (radiance sums, pixels, scatter outputs, camera rays, sphere records...)."""
    ca, cb = canon(a, fold_zero), canon(b, fold_zero)
    if not np.array_equal(ca, cb):
        bad = np.argwhere(ca != cb)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {bad.shape[0]} of {ca.size} values differ; first at {i}: "
                             f"{np.asarray(a, np.float32)[i]!r} vs {np.asarray(b, np.float32)[i]!r}")


def assert_hits_equal(g, o, what="", scene=None):
    """Hit records, every field bit-exact - sphere u,v included: both sides evaluate the same float atan2/asin sequence
    (det_atan2f / det_asinf in oracle/rt_oracle.c and csrc/rt_device.h), so no libm stands between them.
    One exception, and only with `scene` given: records of RECT primitives in the factory form may differ in the sign of a zero
    in t / pos / nrm, because the device skips a RotatedInstance chain whose three angles are 0 (rt_device.h to_leaf_space:
    1*x + (+-0)*y == x for finite input, up to the sign of a zero).  Every other record is compared strictly."""
    assert np.array_equal(g["prim"], o["prim"]), f"{what}: winning primitive differs in {(g['prim'] != o['prim']).sum()} rays"
    assert np.array_equal(g["mat"], o["mat"]), f"{what}: material differs"
    prim = np.asarray(o["prim"])
    rect = np.zeros(prim.shape, bool)
    if scene is not None:
        kind = np.asarray(scene.kind)
        rect[prim >= 0] = kind[prim[prim >= 0]] == 1        # capi.PRIM_RECT
    for f in ("t", "pos", "nrm", "u", "v"):
        ga, oa = np.asarray(g[f]), np.asarray(o[f])
        assert_bits(ga[~rect], oa[~rect], f"{what}.{f}")
        if rect.any():
            assert_bits(ga[rect], oa[rect], f"{what}.{f} (rects)", fold_zero=f in ("t", "pos", "nrm"))


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


# ---- image textures ------------------------------------------------------------------------------------------
def synthetic_texture(w, h, seed, kind="checker"):
    """A deterministic RGBA8 image, (h, w, 4), row 0 = top: a noisy checker board or a colour ramp."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "checker":
        base = np.where(((xx // max(1, w // 8)) + (yy // max(1, h // 8))) % 2 == 0, 200, 60)[..., None] + rng.integers(-40, 40, (h, w, 3))
    else:
        base = np.stack([xx * 255 // max(1, w - 1), yy * 255 // max(1, h - 1), (xx + yy) * 255 // max(1, w + h - 2)], axis=-1) + rng.integers(-8, 8, (h, w, 3))
    img = np.empty((h, w, 4), np.uint8)
    img[..., :3] = np.clip(base, 0, 255).astype(np.uint8)
    img[..., 3] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    return img


def write_tga(path, rgba, kind="rgb24", rle=False, origin="ll", id_bytes=b"", cmap_origin=0):
    """Writes `rgba` ((h, w, 4) uint8, row 0 = top) as a TGA file: kind rgb24 | rgb32 | gray | cmap24 | cmap32,
    optionally run-length encoded, with the first stored pixel at the given corner (ll, ul, lr, ur).
    Returns the RGBA image a decoder must produce (alpha 255 unless the file stores it; grey replicates)."""
    h, w = rgba.shape[:2]
    exp = rgba.copy()
    stored = rgba[::-1] if origin[0] == "l" else rgba
    stored = stored[:, ::-1] if origin[1] == "r" else stored
    desc = (0x20 if origin[0] == "u" else 0) | (0x10 if origin[1] == "r" else 0)
    palette = b""
    cmap_len = cmap_depth = 0
    if kind in ("rgb24", "rgb32"):
        bpp = 3 if kind == "rgb24" else 4
        px = stored[..., [2, 1, 0, 3]][..., :bpp].reshape(-1, bpp)                # B,G,R(,A)
        itype = 2
        if bpp == 3:
            exp[..., 3] = 255
    elif kind == "gray":
        bpp, itype = 1, 3
        px = stored[..., :1].reshape(-1, 1)
        exp[..., 1] = exp[..., 0]; exp[..., 2] = exp[..., 0]; exp[..., 3] = 255
    else:
        bpp, itype = 1, 1
        cbytes = 3 if kind == "cmap24" else 4
        q = (stored[..., :3] // 64).astype(np.uint32)                                 # 64-entry palette from quantised colour
        idx = (q[..., 0] * 16 + q[..., 1] * 4 + q[..., 2]).astype(np.uint8)
        pal = np.array([[(i // 16) * 64 + 17, ((i // 4) % 4) * 64 + 29, (i % 4) * 64 + 5, 100 + i] for i in range(64)], np.uint8)   # R,G,B,A
        palette = pal[:, [2, 1, 0, 3]][:, :cbytes].tobytes()
        cmap_len, cmap_depth = 64, cbytes * 8
        px = (idx + cmap_origin).astype(np.uint8).reshape(-1, 1)                       # stored index = palette entry + first-entry index
        full = pal[idx]
        full = full[:, ::-1] if origin[1] == "r" else full
        full = full[::-1] if origin[0] == "l" else full
        exp = full.copy()
        if cbytes == 3:
            exp[..., 3] = 255
    px = np.ascontiguousarray(px, np.uint8)
    if rle:
        out = bytearray()
        i, n = 0, px.shape[0]
        while i < n:
            run = 1
            while i + run < n and run < 128 and np.array_equal(px[i + run], px[i]):
                run += 1
            if run > 1:
                out.append(0x80 | (run - 1)); out += px[i].tobytes(); i += run
            else:
                j = i + 1
                while j < n and j - i < 128 and not (j + 1 < n and np.array_equal(px[j], px[j + 1])):
                    j += 1
                out.append(j - i - 1); out += px[i:j].tobytes(); i = j
        data = bytes(out); itype += 8
    else:
        data = px.tobytes()
    head = bytes([len(id_bytes), 1 if palette else 0, itype, cmap_origin & 255, cmap_origin >> 8, cmap_len & 255, cmap_len >> 8, cmap_depth, 0, 0, 0, 0,
                  w & 255, w >> 8, h & 255, h >> 8, bpp * 8, desc | (8 if kind == "rgb32" else 0)])
    with open(path, "wb") as f:
        f.write(head + id_bytes + palette + data)
    return exp
