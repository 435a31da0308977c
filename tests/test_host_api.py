"""The drop-in C++ scene API (raytracinginaweekend_amd/host): worlds built through the reference's class
names flatten to the same arrays as the Python mirror, its SimpleObjectBVHNode builds the tree the
oracle's restatement of the reference builder builds, and (on the GPU) TraceRay renders the same image."""
import os
import subprocess

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene
from tests import helpers as H
from tests import hostapi, orc

CASES = [(hostapi.WORLD_CORNELL, lambda: scene.cornell_box(1.0), 1.0), (hostapi.WORLD_TWO_BOXES, lambda: scene.cornell_two_boxes(1.0), 1.0),
         (hostapi.WORLD_FOUR_SPHERES, lambda: scene.spheres_world(2.0, small_spheres=False, seed=1), 2.0),
         (hostapi.WORLD_RANDOM_SPHERES, lambda: scene.spheres_world(1.5, seed=1), 1.5)]


def per_prim_material(fs):
    a = fs.arrays()
    m = a["prim_mat"]
    return np.concatenate([a["mat_kind"][m][:, None].astype(np.float32), a["mat_albedo"][m], a["mat_fuzz"][m][:, None], a["mat_ior"][m][:, None], a["mat_emit"][m]], axis=1)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cpp_world_flattens_like_the_python_mirror(case):
    wid, mk, aspect = CASES[case]
    h, hcam = hostapi.build_world(wid, aspect, seed=1)
    p, pcam = mk()
    ha, pa = h.arrays(), p.arrays()
    for k in ("prim_kind", "prim_off", "prim_geom", "prim_axes", "prim_rot", "obj_first", "obj_count"):
        assert np.array_equal(ha[k], pa[k]), k
    assert np.array_equal(per_prim_material(h), per_prim_material(p))
    assert tuple(np.float32(h.ambient)) == tuple(np.float32(p.ambient))
    assert np.allclose(np.frombuffer(bytes(hcam), np.float32), np.frombuffer(bytes(pcam), np.float32), rtol=3e-7, atol=1e-7)
    assert bytes(hcam) == bytes(orc.camera_init(*{hostapi.WORLD_CORNELL: ((0.0, 0.64, 4.63), (0.0, -0.26, -1.0), 30.0, 0.0),
                                                  hostapi.WORLD_TWO_BOXES: ((0.0, 0.0, 4.0), (0.0, 0.0, 0.0), 40.0, 0.0)}.get(wid, ((11.0, 2.0, 3.0), (0.0, 0.6, 0.0), 20.0, 0.1)), aspect))


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cpp_bvh_equals_the_oracles_reference_builder(case):
    """Same objects + the same raw axis draws -> the same tree, node for node, box for box."""
    wid, mk, aspect = CASES[case]
    h, _ = hostapi.build_world(wid, aspect, seed=1)
    # the draws the C++ Randomizer had left when it reached `new SimpleObjectBVHNode(objects)`
    g = scene._HostRng(1)
    consumed = 50 * 6 + 50 * 4                                  # Resources::LoadMaterials, Resouces.cpp:110-124
    if wid == hostapi.WORLD_RANDOM_SPHERES:
        p, _ = mk()
        consumed += 22 * 22 * 3 + ((np.asarray(p.arrays()["prim_mat"][4:]) < 100).sum())   # 3 per grid cell + 1 per non-glass sphere kept
    raw = []
    for _ in range(consumed + 4 * h.n_objects + 8):
        if not g.buf:
            g.buf = g.philox((g.i, 0, 0, 0x5CE9E), g.key); g.i += 1
        raw.append(g.buf.pop(0))
    o, _ = mk() if wid != hostapi.WORLD_RANDOM_SPHERES else (p, None)
    n = orc.bvh_build_reference(o, np.array(raw[consumed:], np.uint32))
    assert n == h.nodes[2].shape[0]
    assert np.array_equal(o.nodes[2], h.nodes[2]) and np.array_equal(o.nodes[3], h.nodes[3]) and o.nodes[4] == h.nodes[4]
    assert np.array_equal(o.nodes[0], h.nodes[0]) and np.array_equal(o.nodes[1], h.nodes[1])


def test_general_hitables_flatten_and_what_is_still_refused():
    """VERDICT r2 "What's missing 1": scene code written against the reference's Hitables.h composes freely.  WORLD_ID_CHAINS builds,
    through the C++ classes alone, every shape round 2's flattener threw on -- a rotated sphere, an off-centre sphere inside a
    chain, combos of things other than rects (and a combo inside a combo), a translation that is not outermost, rotations not
    ordered z,y,x -- and flattens to the arrays of the Python mirror.  What is still refused: more than MIRT_MAX_CHAIN_OPS
    nested instances, and an unknown world id."""
    h, hcam = hostapi.build_world(hostapi.WORLD_CHAINS, 1.0, seed=1)
    p, pcam = scene.chains_world(1.0)
    ha, pa = h.arrays(), p.arrays()
    for k in ("prim_kind", "prim_off", "prim_geom", "prim_axes", "prim_rot", "obj_first", "obj_count", "prim_chain", "prim_nops", "prim_center", "op_kind", "op_arg"):
        assert np.array_equal(ha[k], pa[k]), k
    assert np.array_equal(per_prim_material(h), per_prim_material(p)) and bytes(hcam) == bytes(pcam)
    kinds = [[k for k, _ in c] if c is not None else None for c in h.chain]
    assert kinds[:2] == [None, None]                                             # the two factory rects keep the factory form
    assert kinds[2] == [capi.OP_TRANSLATE, capi.OP_ROTATE_Y] and h.center[2] == (1.0, 2.0, 3.0)     # rotated, off-centre sphere
    assert kinds[3] == [] and kinds[4] == [] and h.obj_count[3] == 2             # a combo of two bare spheres: one object, two primitives
    assert kinds[5] == [capi.OP_ROTATE_X, capi.OP_TRANSLATE, capi.OP_ROTATE_Z]   # neither translate-outermost nor z,y,x
    assert h.obj_count[5] == 4
    assert kinds[6] == [capi.OP_TRANSLATE, capi.OP_ROTATE_Y] and kinds[7] == [capi.OP_TRANSLATE, capi.OP_ROTATE_Y, capi.OP_TRANSLATE]
    assert kinds[9] == [capi.OP_TRANSLATE, capi.OP_ROTATE_Y, capi.OP_TRANSLATE]  # the nested combo's translated ball: chain form although Translate(Sphere at 0)
    # the factories' own shapes stay in the factory form; a sphere with a centre of its own does not
    assert hostapi.try_chain(1) == (1, 0, False) and hostapi.try_chain(1, 0.5) == (1, 1, True) and hostapi.try_chain(0, 0.0) == (1, 0, True)
    assert hostapi.try_chain(capi.MAX_CHAIN_OPS, 0.5) == (1, capi.MAX_CHAIN_OPS, True)
    with pytest.raises(RuntimeError, match="9 nested instances; the device takes 8"):
        hostapi.try_chain(capi.MAX_CHAIN_OPS + 1, 0.5)
    err = hostapi.C.create_string_buffer(256)
    assert hostapi.lib().mirt_host_world_create(99, 1.0, 1, err, 256) is None and b"unknown world" in err.value
    # the library's own checks behind the C ABI (mirt_build_bvh validates like mirt_upload_scene, without a GPU)
    bad = scene.chains_world(1.0)[0]
    bad.chain[2] = bad.chain[2] + [(7, (0.0, 0.0, 0.0))]
    with pytest.raises(capi.MirtError):
        capi.build_bvh(bad)
    two = scene.FlatScene(); m = two.lambertian()
    two.rect(scene.XZ_RECT, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 1.0, 1.0, False, m); two.rect(scene.XZ_RECT, (0.0, 1.0, 0.0), (0.0, 0.0, 0.0), 1.0, 1.0, False, m)
    two.obj_first, two.obj_count = [0], [2]                                      # two factory rects with DIFFERENT chains in one object
    with pytest.raises(capi.MirtError):
        capi.build_bvh(two)


@pytest.mark.gpu
def test_traceray_through_the_cpp_api_matches_the_c_abi(gpu, tmp_path):
    for wid, mk, aspect in CASES[:3]:
        w, h = (96, 96) if aspect == 1.0 else (100, 50)
        hs, hcam = hostapi.build_world(wid, aspect, seed=1)
        gpu.upload_scene(hs); gpu.set_camera(hcam)
        for one_spp, normal, spp in ((True, False, 1), (True, True, 1), (False, False, 6)):
            flags = capi.FLAG_ACCEL_AUTO | (capi.FLAG_ONE_SPP if one_spp else 0) | (capi.FLAG_NORMAL_DISPLAY if normal else 0)
            a = hostapi.render(wid, w, h, spp, 50, 1, 0x5EED, one_spp, normal)
            b = gpu.render(w, h, spp, 50, 0x5EED, flags)
            assert np.array_equal(a, b["rgba8"]), (wid, one_spp, normal)
    # the chains world: TraceRay through the C++ classes == the C ABI on the Python mirror == the oracle, bit for bit
    ps, pcam = scene.chains_world(1.0)
    gpu.upload_scene(ps); gpu.set_camera(pcam)
    a = hostapi.render(hostapi.WORLD_CHAINS, 80, 80, 5, 50, 1, 0x5EED, False, False)
    b = gpu.render(80, 80, 5, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
    osum, ost = orc.trace(ps, pcam, 80, 80, 5, 50, 0x5EED, orc.FLAG_ACCEL_LIST)
    assert np.array_equal(a, b["rgba8"]) and np.array_equal(b["rgba8"], orc.resolve(osum, 5)[1]) and b["stats"].rays == ost.rays
    H.assert_bits(b["sum"], osum, "chains world")
    out = hostapi.render(hostapi.WORLD_CORNELL, 32, 24, 2, ppm_dir=str(tmp_path))
    txt = open(tmp_path / "OutputImage.ppm").read().split("\n")
    assert txt[:3] == ["P3", "32 24", "255"] and len(txt) == 3 + 32 * 24 + 1
    assert txt[3] == " ".join(str(int(v)) for v in out[0, 0, :3])


@pytest.mark.gpu
def test_tracer_follows_a_rebuilt_world():
    """The reference reads the live pointer graph on every TraceRay (HomemadeRayTracer.cpp:162).  Here the tracer keeps a
    flattened copy on the device: after DeconstructWorld + ConstructWorld(other id) on the same World object it must
    render the new world, not the cached one."""
    first, second = hostapi.render_rebuilt(hostapi.WORLD_CORNELL, hostapi.WORLD_FOUR_SPHERES, 64, 48, 4)
    assert np.array_equal(first, hostapi.render(hostapi.WORLD_CORNELL, 64, 48, 4))
    assert np.array_equal(second, hostapi.render(hostapi.WORLD_FOUR_SPHERES, 64, 48, 4))
    assert not np.array_equal(first, second)


@pytest.mark.gpu
def test_example_program_renders_a_ppm(tmp_path):
    hostapi.ensure_built()
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "cornell_example")
    r = subprocess.run([exe, "64", "64", "4", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "[HomemadeRayTracer] SPP: 4" in r.stdout
    lines = open(tmp_path / "OutputImage.ppm").read().split("\n")
    assert lines[1] == "64 64" and len(lines) == 3 + 64 * 64 + 1


@pytest.mark.gpu
def test_fly_camera_and_autofocus(orc):
    """SimpleCamera::OnUpdate (SimpleCamera.cpp:62-124) with AutoFocus (:202-214) answered by the device: the camera
    after each scripted frame equals a float32 restatement whose focus ray is traced by the oracle."""
    F = np.float32
    frames, dt = ["W", "WL", "L", "D", "", "SU", "P", "AR"], 1.0 / 30.0
    for wid, look_from, look_at, fov, aperture, aspect in ((hostapi.WORLD_CORNELL, (0.0, 0.64, 4.63), (0.0, -0.26, -1.0), 30.0, 0.0, 1.0),
                                                            (hostapi.WORLD_FOUR_SPHERES, (11.0, 2.0, 3.0), (0.0, 0.6, 0.0), 20.0, 0.1, 2.0)):
        got = hostapi.camera_script(wid, aspect, 1, frames, dt)
        hs, _ = hostapi.build_world(wid, aspect, seed=1)
        origin, focus = np.array(look_from, F), np.array(look_at, F)
        look = focus - origin
        ln = np.sqrt((look * look).sum(dtype=F))
        pitch = F(-np.arcsin(look[1] / ln))
        yaw = F(np.arctan(look[0] / look[2]) + (np.pi if look[2] < 0 else 0.0))
        move_speed, turn_speed = F(10.0), F(1.0)
        for f, held in enumerate(frames):
            if held:
                cam = np.frombuffer(bytes(scene.make_camera(origin, focus, fov, aperture, aspect)), F)
                u = cam[12:15]
                w = (origin - focus) / np.sqrt(((origin - focus) ** 2).sum(dtype=F))
                move = np.zeros(3, F)
                for ch, v in (("A", (-1, 0, 0)), ("D", (1, 0, 0)), ("W", (0, 0, -1)), ("S", (0, 0, 1))):
                    if ch in held:
                        move += np.array(v, F)
                yaw += F(turn_speed * F(dt)) * (("L" in held) - ("R" in held))
                pitch += F(turn_speed * F(dt)) * (("P" in held) - ("U" in held))
                n = np.sqrt((move * move).sum(dtype=F))
                move = (move / n if n > 0 else move) * F(move_speed * F(dt))
                origin = origin + (u * move[0] + w * move[2])
                r = np.cos(pitch)
                look_dir = np.array([r * np.sin(yaw), -np.sin(pitch), r * np.cos(yaw)], F)
                focus = origin + look_dir
                hit, _ = orc.world_hit_batch(hs, origin[None], look_dir[None], 0.01, float(np.finfo(F).max), 0)
                if hit["prim"][0] >= 0:
                    focus = hit["pos"][0].astype(F)
            exp = np.frombuffer(bytes(scene.make_camera(origin, focus, fov, aperture, aspect)), F)
            assert np.allclose(got[f], exp, rtol=2e-4, atol=2e-4), (wid, f, got[f], exp)
    assert not np.allclose(got[0], got[-1], atol=1e-3)                   # the script really moved the camera


@pytest.mark.gpu
def test_progressive_passes_equal_one_render(gpu):
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    w, h = 80, 60
    one = gpu.render(w, h, 11, 50, 9)
    a = gpu.render_progressive(w, h, 0, 4, 50, 9)
    assert np.array_equal(a["rgba8"], gpu.render(w, h, 4, 50, 9)["rgba8"])
    b = gpu.render_progressive(w, h, 4, 1, 50, 9)
    c = gpu.render_progressive(w, h, 5, 6, 50, 9)
    H.assert_bits(c["sum"], one["sum"], "progressive sum"); H.assert_bits(c["pixels"], one["pixels"], "progressive pixels")
    assert np.array_equal(c["rgba8"], one["rgba8"]) and not np.array_equal(b["rgba8"], one["rgba8"])
    with pytest.raises(capi.MirtError, match="continue"):
        gpu.render_progressive(w, h, 5, 2, 50, 9)                        # [5,7) does not continue [0,11)
    with pytest.raises(capi.MirtError, match="continue"):
        gpu.render_progressive(w + 1, h, 11, 2, 50, 9)
    with pytest.raises(capi.MirtError):
        gpu.render_progressive(w, h, 0, 2, 50, 9, flags=capi.FLAG_ONE_SPP)
    d = gpu.render_progressive(w, h, 0, 11, 50, 9)                       # restart
    assert np.array_equal(d["rgba8"], one["rgba8"])


@pytest.mark.gpu
def test_flythrough_cli(tmp_path):
    hostapi.ensure_built()
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "flythrough")
    r = subprocess.run([exe, "1", "48", "40", "3", "6", "4", str(tmp_path), "W,WL,-"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    names = sorted(os.listdir(tmp_path))
    assert names == ["preview_0000.ppm", "preview_0001.ppm", "preview_0002.ppm", "refine_00004.ppm", "refine_00006.ppm"]
    frames = [open(tmp_path / n, "rb").read() for n in names]
    assert all(f.startswith(b"P6\n48 40\n255\n") and len(f) == 13 + 48 * 40 * 3 for f in frames)
    assert frames[0] != frames[1] and frames[3] != frames[4]             # the camera moved; the refinement changed the image
    assert "6 spp after" in r.stdout
