"""Image textures: SimpleTexture2D_TGAImage::Sample (SimpleTexture2D.cpp:130-147), the TGA decoder that
replaces tga_reader.cpp, CalculateUV (Hitables.cpp:62-78) made live.  The reference ships no world that uses them, so the
scenes here are the host layer's WORLD_ID_TEXTURED and synthetic images written by tests/helpers.write_tga."""
import itertools
import os

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene
from tests import helpers as H
from tests import hostapi

F = np.float32


def ref_sample(tex, uv):
    """numpy restatement of Sample: nearest texel, MSVC float->UINT32, clamp, /255.0f."""
    h, w = tex.shape[:2]
    u, v = uv[:, 0].astype(F), uv[:, 1].astype(F)

    def to_u32(f):
        f64 = f.astype(np.float64)
        ok = np.isfinite(f64) & (np.abs(f64) < 2.0 ** 63)
        t = np.where(ok, np.trunc(np.where(ok, f64, 0.0)), 0.0).astype(np.int64)
        return (t & 0xFFFFFFFF).astype(np.uint64)
    with np.errstate(all="ignore"):
        i = to_u32(u * F(w)); j = to_u32((F(1.0) - v) * F(h) - F(0.001))
    i = np.minimum(i, w - 1).astype(np.int64); j = np.minimum(j, h - 1).astype(np.int64)
    return (tex[j, i, :3].astype(F) / F(255.0)).astype(F)


def edge_uv(n, seed):
    rng = np.random.default_rng(seed)
    uv = rng.uniform(0, 1, (n, 2)).astype(F)
    special = np.array([[0, 0], [1, 1], [0, 1], [1, 0], [0.999999, 1e-7], [-0.25, 0.5], [1.5, -0.5], [0.5, 2.0], [-3.0, 0.5],
                        [np.nan, 0.5], [0.5, np.nan], [np.inf, 0.5], [0.5, -np.inf], [1e30, 1e-30]], F)
    uv[: len(special)] = special
    return uv


def textured_scene():
    soil, checker = H.synthetic_texture(37, 23, 1, "ramp"), H.synthetic_texture(64, 48, 2, "checker")
    return scene.textured_world(1.5, soil, checker), (soil, checker)


def test_oracle_sample_known_answers(orc):
    (s, _), (soil, checker) = textured_scene()
    for t, tex in enumerate((soil, checker)):
        uv = edge_uv(20000, 10 + t)
        H.assert_bits(orc.tex_sample_batch(s, t, uv), ref_sample(tex, uv), f"texture {t}")
    one = scene.FlatScene(); one.texture(np.array([[[10, 20, 30, 40]]], np.uint8)); one.lambertian(texture=0)
    one.sphere((0, 0, 0), 1.0, 0)
    assert np.array_equal(orc.tex_sample_batch(one, 0, edge_uv(64, 3)), np.tile((np.array([10, 20, 30], F) / F(255.0))[None], (64, 1)))


def test_tga_decoder_id_field_and_colormap_origin(tmp_path):
    """Pins a deliberate deviation from the reference's vendored reader (tga_reader.cpp): that one never skips the image-ID
    field (its palette and pixels start at byte 18 whatever header byte 0 says, tga_reader.cpp:58,75-83) and evaluates
    `byte & 0xFF - colormapOrigin` as `byte & (0xFF - origin)` (tga_reader.cpp:156).  DecodeTGA follows the TGA
    specification instead -- ID field skipped, index minus first-entry index -- so files with neither (the reference's own
    asset among them) decode to the same texels and legal files with either decode correctly rather than to noise."""
    img = H.synthetic_texture(13, 9, 11, "checker")
    for kind, rle, idb, org in (("rgb24", False, b"an image id", 0), ("gray", False, b"\x00\x01\x02", 0), ("cmap24", False, b"", 7),
                                ("cmap32", True, b"xy", 100), ("cmap24", True, b"", 191)):
        path = str(tmp_path / f"{kind}{int(rle)}{len(idb)}{org}.tga")
        exp = H.write_tga(path, img, kind, rle, "ul", id_bytes=idb, cmap_origin=org)
        assert np.array_equal(hostapi.decode_tga(path), exp), (kind, rle, idb, org)


@pytest.mark.parametrize("kind,rle,origin", [(k, r, o) for k, r, o in itertools.product(("rgb24", "rgb32", "gray", "cmap24", "cmap32"), (False, True), ("ll", "ul", "lr", "ur"))])
def test_tga_decoder(tmp_path, kind, rle, origin):
    img = H.synthetic_texture(19, 11, 5, "checker")
    img[3, 2:9] = img[3, 2]                                      # a run for the RLE encoder
    exp = H.write_tga(str(tmp_path / "t.tga"), img, kind, rle, origin, id_bytes=b"id!" if rle else b"")
    got = hostapi.decode_tga(str(tmp_path / "t.tga"))
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_tga_decoder_rejects_bad_files(tmp_path):
    img = H.synthetic_texture(8, 8, 6)
    H.write_tga(str(tmp_path / "ok.tga"), img)
    raw = open(tmp_path / "ok.tga", "rb").read()
    for name, data in (("short", raw[:10]), ("cut", raw[:-20]), ("type", raw[:2] + bytes([7]) + raw[3:]), ("depth", raw[:16] + bytes([16]) + raw[17:]),
                       ("empty", raw[:12] + bytes([0, 0]) + raw[14:])):
        open(tmp_path / f"{name}.tga", "wb").write(data)
        with pytest.raises(RuntimeError, match="DecodeTGA"):
            hostapi.decode_tga(str(tmp_path / f"{name}.tga"))
    with pytest.raises(RuntimeError, match="cannot open"):
        hostapi.decode_tga(str(tmp_path / "missing.tga"))


def test_reference_asset_decodes():
    """The one image asset present in the reference checkout (the other is missing there, .MISSING_LARGE_BLOBS)."""
    p = "/root/reference/RayTracer/Assets/pro_metal_checker_plate_001_c.tga"
    if not os.path.exists(p):
        pytest.skip("reference checkout not present")
    img = hostapi.decode_tga(p, cap=1 << 26)
    assert img.shape[0] >= 256 and img.shape[1] >= 256 and img[..., :3].std() > 5


def test_host_textured_world_flattens_like_the_python_mirror(tmp_path, monkeypatch):
    (p, pcam), (soil, checker) = textured_scene()
    H.write_tga(str(tmp_path / "pab_ground_soil_001_c.tga"), soil, "rgb24", True, "ll")
    H.write_tga(str(tmp_path / "pro_metal_checker_plate_001_c.tga"), checker, "rgb32", False, "ul")
    monkeypatch.setenv("MIRT_ASSETS", str(tmp_path))
    h, hcam = hostapi.build_world(hostapi.WORLD_TEXTURED, 1.5, seed=1)
    ha, pa = h.arrays(), p.arrays()
    for k in ("prim_kind", "prim_off", "prim_geom", "prim_axes", "prim_rot", "obj_first", "obj_count", "tex_width", "tex_height", "tex_offset"):
        assert np.array_equal(ha[k], pa[k]), k
    soil24 = soil.copy(); soil24[..., 3] = 255                      # the 24-bit file has no alpha
    assert np.array_equal(h.textures[0], soil24) and np.array_equal(h.textures[1], checker)
    assert np.array_equal(ha["mat_texture"][ha["prim_mat"]], pa["mat_texture"][pa["prim_mat"]])
    assert np.allclose(np.frombuffer(bytes(hcam), F), np.frombuffer(bytes(pcam), F), rtol=3e-7, atol=1e-7)
    monkeypatch.setenv("MIRT_ASSETS", str(tmp_path / "nowhere"))     # files absent: mid-grey stand-ins, no textures
    g, _ = hostapi.build_world(hostapi.WORLD_TEXTURED, 1.5, seed=1)
    assert not g.textures and np.array_equal(g.arrays()["mat_albedo"][g.arrays()["prim_mat"][0]], np.array([0.5, 0.5, 0.5], F))


def test_oracle_textured_render_uses_the_image(orc):
    (s, cam), (soil, checker) = textured_scene()
    a, _ = orc.trace(s, cam, 60, 40, 4, 50, 3, orc.FLAG_ACCEL_LIST)
    s2, _ = scene.textured_world(1.5, 255 - soil, checker)
    b, _ = orc.trace(s2, cam, 60, 40, 4, 50, 3, orc.FLAG_ACCEL_LIST)
    assert not np.array_equal(a, b) and np.isfinite(a).all()
    osum, _ = orc.trace(s, cam, 60, 40, 4, 50, 3, 0 if s.nodes else orc.FLAG_ACCEL_LIST)
    assert np.array_equal(osum, a)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_sample_parity(gpu, orc):
    (s, cam), _ = textured_scene()
    gpu.upload_scene(s)
    for t in (0, 1):
        uv = edge_uv(50000, 20 + t)
        H.assert_bits(gpu.tex_sample_batch(t, uv), orc.tex_sample_batch(s, t, uv), f"texture {t}")
    with pytest.raises(capi.MirtError, match="out of range"):
        gpu.tex_sample_batch(2, edge_uv(16, 1))


@pytest.mark.gpu
def test_gpu_textured_scatter_parity(gpu, orc):
    (s, cam), _ = textured_scene()
    gpu.upload_scene(s)
    rng = np.random.default_rng(8)
    n = 8192
    hits = np.zeros(n, capi.HIT_DTYPE)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits["nrm"] = nrm.astype(F); hits["pos"] = rng.uniform(-1, 1, (n, 3)).astype(F); hits["t"] = 1.0
    hits["mat"] = np.arange(n) % len(s.mkind)
    uv = edge_uv(n, 9); hits["u"] = uv[:, 0]; hits["v"] = uv[:, 1]
    d = rng.normal(size=(n, 3)).astype(F)
    draws = rng.integers(0, 2 ** 32, (n, 3 * 64), dtype=np.uint64).astype(np.uint32)
    g = gpu.scatter_batch(hits, d, draws); o = orc.scatter_batch(s, hits, d, draws)
    assert np.array_equal(g["alive"], o["alive"]) and np.array_equal(g["consumed"], o["consumed"])
    for f in ("atten", "org", "dir", "emit"):
        H.assert_bits(g[f], o[f], f"textured scatter.{f}")
    assert len(np.unique(o["atten"][hits["mat"] == 0], axis=0)) > 100           # the Lambertian's albedo really varies with u,v


@pytest.mark.gpu
def test_gpu_textured_image(gpu, orc):
    """Rect u,v are IEEE divisions and sphere u,v the shared det_atan2f/det_asinf sequence, so every texel fetch - and with
    it every path - is the oracle's: the image is compared bit for bit, in the list walk against the reference-order oracle
    and in the library's near-first walk against its restatement."""
    (s, cam), _ = textured_scene()
    gpu.upload_scene(s); gpu.set_camera(cam)
    w, h = 120, 80
    for flags, oflags in ((capi.FLAG_ACCEL_LIST, orc.FLAG_ACCEL_LIST), (capi.FLAG_ACCEL_ORDERED, orc.FLAG_ACCEL_ORDERED)):
        if flags == capi.FLAG_ACCEL_ORDERED:
            capi.build_bvh(s); gpu.upload_scene(s); gpu.set_camera(cam)           # both sides walk this tree
        g = gpu.render(w, h, 8, 50, 21, flags)
        osum, ost = orc.trace(s, cam, w, h, 8, 50, 21, oflags)
        H.assert_bits(g["sum"], osum, f"textured image flags={flags}")
        assert int(g["stats"].rays) == int(ost.rays)
    g = gpu.render(w, h, 8, 50, 21, 0); osum, _ = orc.trace(s, cam, w, h, 8, 50, 21, orc.FLAG_ACCEL_LIST)
    assert H.psnr(g["sum"], osum) > 45                                            # default flags: same picture whichever walk the library picks
    with pytest.raises(capi.MirtError, match="single-colour"):
        gpu.set_option("engine", 0)
        try:
            gpu.render(w, h, 2, 50, 21, capi.FLAG_ACCEL_LIST)
        finally:
            gpu.set_option("engine", 1)


@pytest.mark.gpu
def test_gpu_host_api_textured_world(gpu, tmp_path, monkeypatch):
    (p, pcam), (soil, checker) = textured_scene()
    H.write_tga(str(tmp_path / "pab_ground_soil_001_c.tga"), soil, "rgb32", False, "ll")
    H.write_tga(str(tmp_path / "pro_metal_checker_plate_001_c.tga"), checker, "rgb32", True, "ur")
    monkeypatch.setenv("MIRT_ASSETS", str(tmp_path))
    a = hostapi.render(hostapi.WORLD_TEXTURED, 90, 60, 5, 50, 1, 0x5EED, False, False)
    gpu.upload_scene(p); gpu.set_camera(pcam)
    b = gpu.render(90, 60, 5, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
    assert np.array_equal(a, b["rgba8"])                                         # same device code either way: identical


# ------------------------------------------------------------------------------------------------------------ golden
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f7_textures.npz")


def golden_scene():
    g = np.load(GOLD)
    assert np.array_equal(g["soil"], H.synthetic_texture(37, 23, 1, "ramp")) and np.array_equal(g["checker"], H.synthetic_texture(64, 48, 2, "checker"))
    return g, scene.textured_world(1.5, g["soil"], g["checker"])


def test_oracle_matches_texture_fixture(orc):
    g, (s, cam) = golden_scene()
    H.assert_bits(orc.tex_sample_batch(s, 0, g["uv"]), g["sample0"], "sample0"); H.assert_bits(orc.tex_sample_batch(s, 1, g["uv"]), g["sample1"], "sample1")
    o = orc.scatter_batch(s, g["hits"], g["dir_in"], g["draws"])
    for k in ("atten", "org", "dir", "emit"):
        H.assert_bits(o[k], g["out_" + k], "scatter." + k)
    osum, _ = orc.trace(s, cam, 60, 40, 4, 50, 3, orc.FLAG_ACCEL_LIST)
    H.assert_bits(osum, g["image_sum"], "image")


@pytest.mark.gpu
def test_gpu_matches_texture_fixture(gpu):
    g, (s, cam) = golden_scene()
    gpu.upload_scene(s); gpu.set_camera(cam)
    H.assert_bits(gpu.tex_sample_batch(0, g["uv"]), g["sample0"], "sample0"); H.assert_bits(gpu.tex_sample_batch(1, g["uv"]), g["sample1"], "sample1")
    o = gpu.scatter_batch(g["hits"], g["dir_in"], g["draws"])
    assert np.array_equal(o["alive"], g["out_alive"]) and np.array_equal(o["consumed"], g["out_consumed"])
    for k in ("atten", "org", "dir", "emit"):
        H.assert_bits(o[k], g["out_" + k], "scatter." + k)
    img = gpu.render(60, 40, 4, 50, 3, capi.FLAG_ACCEL_LIST)["sum"]
    assert (np.abs(img - g["image_sum"]).max(axis=2) > 0).mean() < 0.02 and H.psnr(img, g["image_sum"]) > 45
