"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mirt.h declares,
its structs have the layout the Python mirror assumes, and it fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mirt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libmirt.so does not export {n}"
    assert sorted(capi.EXPORTS) == names


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mirt.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mirt_scene), '
                   'sizeof(mirt_camera), sizeof(mirt_hit), sizeof(mirt_render_params), sizeof(mirt_stats), offsetof(mirt_scene, ambient));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out == [C.sizeof(capi.SceneStruct), C.sizeof(capi.CameraStruct), capi.HIT_DTYPE.itemsize, C.sizeof(capi.RenderParams),
                   C.sizeof(capi.Stats), capi.SceneStruct.ambient.offset]
    # the oracle's orc_scene / orc_hit are layout-identical by construction: same check against its header
    src2 = tmp_path / "sz2.c"
    src2.write_text('#include <stdio.h>\n#include "rt_oracle.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(orc_scene), sizeof(orc_camera), '
                    'sizeof(orc_hit), offsetof(orc_scene, ambient));return 0;}\n')
    exe2 = tmp_path / "sz2"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "oracle"), str(src2), "-o", str(exe2)], check=True)
    out2 = [int(x) for x in subprocess.run([str(exe2)], check=True, capture_output=True, text=True).stdout.split()]
    assert out2 == [out[0], out[1], out[2], out[5]]


def test_abi_guard_rejects_a_host_built_against_another_header():
    """include/mirt.h "ABI guard": a host with a shorter mirt_stats (the abort of round 1: a stale libmirt_host.so handed a
    grown libmirt.so a 512-byte-shorter struct on its stack) gets an error code at load and per call, never a write."""
    lib = capi.load()
    sizes = [C.sizeof(capi.SceneStruct), C.sizeof(capi.CameraStruct), capi.HIT_DTYPE.itemsize, C.sizeof(capi.RenderParams), C.sizeof(capi.Stats)]
    assert lib.mirt_abi_version() == capi.ABI_VERSION and lib.mirt_check_abi(capi.ABI_VERSION, *sizes) == 0
    assert lib.mirt_check_abi(capi.ABI_VERSION - 1, *sizes) == -1
    short = sizes[:4] + [sizes[4] - 512]
    assert lib.mirt_check_abi(capi.ABI_VERSION, *short) == -1
    assert b"stats" in lib.mirt_last_error(None)
    # per call, host-only entry point: a scene struct that claims another size is refused before any pointer in it is followed
    s, _ = scene.cornell_box(1.0)
    st = s.as_struct()
    st.struct_size -= 8
    out = np.zeros(64, np.float32)
    o = lambda: out.ctypes.data_as(C.c_void_p)
    root, n = C.c_int32(), C.c_uint32()
    assert lib.mirt_build_bvh(C.byref(st), 8, o(), o(), o(), o(), C.byref(root), C.byref(n), None) == -1
    assert b"struct_size" in lib.mirt_last_error(None) and not out.any()


@pytest.mark.skipif(capi.load().mirt_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_a_loud_failure_not_a_fallback():
    with pytest.raises(capi.MirtError, match="no HIP device"):
        capi.Context(0)


def test_product_code_never_touches_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracinginaweekend_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".c")) or f == "Makefile":
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"rt_oracle|libmirt_oracle|from tests|import tests|oracle/", t):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_flat_scene_shapes():
    s, cam = scene.cornell_box(1.0)
    a = s.arrays()
    assert s.n_prims == 9 and s.n_objects == 9 and a["prim_geom"].shape == (9, 5) and a["mat_albedo"].shape == (7, 3)
    assert a["prim_axes"][0] == capi.axes_pack(0, 2, 1, 1, 1)                  # XZ light, reversed, rotation chain present
    assert cam.lens_radius == 0.0
    s2, cam2 = scene.cornell_two_boxes(1.0)
    assert s2.n_prims == 18 and s2.n_objects == 8 and s2.obj_count[-1] == 6
    s3, cam3 = scene.spheres_world(1.5)
    assert 470 < s3.n_objects < 489 and abs(cam3.lens_radius - 0.05) < 1e-9     # SURVEY: ~486 spheres
    s4, _ = scene.spheres_world(2.0, small_spheres=False)
    assert s4.n_objects == 4
    assert np.array_equal(scene.spheres_world(1.5)[0].arrays()["prim_off"], a3 := s3.arrays()["prim_off"]) and a3.shape[1] == 3   # reproducible


def test_spp_ranges_tile_the_sample_space():
    for world in (1, 2, 3, 4, 8):
        for spp in (0, 1, 7, 100, 1024):
            r = [sharding.spp_range(k, world, spp) for k in range(world)]
            assert r[0][0] == 0 and sum(n for _, n in r) == spp
            for (b0, n0), (b1, _) in zip(r, r[1:]):
                assert b0 + n0 == b1
    assert [sharding.spp_range(k, 8, 1024) for k in (0, 7)] == [(0, 128), (896, 128)]


def test_own_bvh_builder_makes_valid_trees():
    """mirt_build_bvh (host-only): every object is a leaf exactly once, every box contains its children, and the
    oracle finds the same closest t through it as through the list."""
    from tests import orc
    from tests import helpers as H
    for mk in (lambda: scene.cornell_box(1.0), lambda: scene.cornell_two_boxes(1.0), lambda: scene.spheres_world(1.5)):
        s, cam = mk()
        depth = capi.build_bvh(s)
        mn, mx, l, r, root = s.nodes
        omn, omx = orc.object_bounds(s)
        seen = []
        def walk(c, lvl):
            if c == capi.CHILD_NONE:
                return None
            if c < 0:
                seen.append(~c); return omn[~c], omx[~c], lvl
            boxes = [b for b in (walk(l[c], lvl + 1), walk(r[c], lvl + 1)) if b is not None]
            for bmn, bmx, _ in boxes:
                assert (mn[c] <= bmn).all() and (mx[c] >= bmx).all()
            return mn[c], mx[c], max(b[2] for b in boxes)
        assert walk(root, 1)[2] - 1 == depth
        assert sorted(seen) == list(range(s.n_objects))
        org, d = H.ray_batch(4096, 5, extent=6.0)
        a, sa = orc.world_hit_batch(s, org, d, 0.001, 3.0e38, 0)
        b, sb = orc.world_hit_batch(s, org, d, 0.001, 3.0e38, orc.FLAG_ACCEL_LIST)
        H.assert_bits(a["t"], b["t"], "own tree vs list")
        if s.n_objects > 100:
            assert sa.prims_tested < 0.05 * sb.prims_tested            # the tree prunes


def test_image_writers(tmp_path):
    """P3 is the reference's format (PPMImageMaker.cpp:31-43); P6 and PFM carry the same pixels."""
    rng = np.random.default_rng(3)
    w, h = 7, 5
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    flt = rng.normal(size=(h, w, 3)).astype(np.float32)
    p3, p6, pfm = (str(tmp_path / n) for n in ("a.ppm", "b.ppm", "c.pfm"))
    capi.write_image(p3, capi.IMG_P3, rgba8=rgba); capi.write_image(p6, capi.IMG_P6, rgba8=rgba); capi.write_image(pfm, capi.IMG_PFM, rgb_float=flt)
    tok = open(p3).read().split()
    assert tok[:4] == ["P3", str(w), str(h), "255"] and np.array_equal(np.array(tok[4:], np.uint8).reshape(h, w, 3), rgba[..., :3])
    raw = open(p6, "rb").read()
    head = f"P6\n{w} {h}\n255\n".encode()
    assert raw.startswith(head) and np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3), rgba[..., :3])
    raw = open(pfm, "rb").read()
    head = f"PF\n{w} {h}\n-1.0\n".encode()
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(h, w, 3)[::-1], flt)      # PFM stores the bottom row first
    with pytest.raises(capi.MirtError):
        capi.write_image(str(tmp_path / "d.pfm"), capi.IMG_PFM, rgba8=rgba)                        # PFM needs floats
    with pytest.raises(capi.MirtError):
        capi.write_image(str(tmp_path / "nodir" / "e.ppm"), capi.IMG_P6, rgba8=rgba)


def _sanitized(tmp_path, name, sources, extra):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe] + sources + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_scene_validation_under_sanitizers(tmp_path):
    """Host code that reads caller-supplied arrays (validate, prepare_tree, the SAH builder) on random scenes with corrupted
    trees, object runs and material indices, built with ASan + UBSan: every scene is rejected or yields a valid tree."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = _sanitized(tmp_path, "fuzz_scene", [os.path.join(root, "tests", "fuzz", "fuzz_scene.cpp")], ["-I" + os.path.join(root, "raytracinginaweekend_amd", "csrc")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "accepted" in r.stdout, r.stdout[-500:] + r.stderr[-3000:]


def test_tga_decoder_under_sanitizers(tmp_path):
    """DecodeTGA on 20,000 mutations of valid files (all kinds, RLE, origins), ASan + UBSan: decodes or throws, never overruns."""
    import itertools
    from tests import helpers as H
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "raytracinginaweekend_amd", "host")
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = _sanitized(tmp_path, "fuzz_tga", [os.path.join(root, "tests", "fuzz", "fuzz_tga.cpp"), os.path.join(host, "host.cpp")],
                     ["-I" + host, "-L" + libdir, "-lmirt", "-Wl,-rpath," + libdir])
    img = H.synthetic_texture(19, 11, 5, "checker")
    seeds = []
    for i, (k, rle, o) in enumerate(itertools.product(("rgb24", "rgb32", "gray", "cmap24", "cmap32"), (False, True), ("ll", "ur"))):
        seeds.append(str(tmp_path / f"s{i}.tga")); H.write_tga(seeds[-1], img, k, rle, o)
    r = subprocess.run([exe] + seeds, capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "decoded" in r.stdout, r.stdout[-500:] + r.stderr[-3000:]
