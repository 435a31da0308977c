"""mirt_multi: one render sharded by samples per pixel over several GPUs inside one process (the reference's only
parallelism is the OpenMP row loop of HomemadeRayTracer.cpp:107-110).  A one-GPU box exercises everything but the xGMI hop by
listing device 0 more than once (peer reduce); the RCCL reduce runs there as a one-rank communicator; with two or more GPUs the
real thing runs too."""
import ctypes as C

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene, sharding
from tests import helpers as H


def test_native_spp_ranges_equal_the_python_ones():
    for world in (1, 2, 3, 5, 8, 16):
        for begin, count in ((0, 0), (0, 1), (0, 7), (3, 100), (0, 1024), (17, 16384)):
            got = [capi.multi_spp_range(r, world, begin, count) for r in range(world)]
            want = [(begin + b, n) for b, n in (sharding.spp_range(r, world, count) for r in range(world))]
            assert got == want


@pytest.mark.skipif(capi.load().mirt_device_count() > 0, reason="a GPU is present")
def test_multi_without_a_gpu_fails_loudly():
    with pytest.raises(capi.MirtError, match="no HIP device"):
        capi.MultiContext([0, 0])
    with pytest.raises(capi.MirtError, match="device ids"):
        capi.MultiContext([])


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["cornell", "spheres"])
def test_two_and_three_shards_on_one_gpu(gpu, world):
    """Peer reduce with device 0 listed G times: the G partial sums are exactly what G single renders of the ranges give, added
    in rank order -- so the sharded image is predicted bit for bit, and it equals the one-device image up to re-association."""
    s, cam = scene.cornell_box(1.0) if world == "cornell" else scene.spheres_world(1.5)
    w, h, spp = (96, 96, 13) if world == "cornell" else (120, 80, 7)
    gpu.upload_scene(s); gpu.set_camera(cam)
    one = gpu.render(w, h, spp, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
    for G in (2, 3):
        with capi.MultiContext([0] * G) as m:
            m.upload_scene(s); m.set_camera(cam)
            r = m.render(w, h, spp, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
            again = m.render(w, h, spp, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
        info = r["info"]
        assert info.n_devices == G and info.reduce_kind == capi.REDUCE_PEER and info.reduce_bytes == w * h * 12
        assert [(info.spp_begin[i], info.spp_count[i]) for i in range(G)] == [sharding.spp_range(i, G, spp) for i in range(G)]
        expect = None
        for i in range(G):
            b, n = sharding.spp_range(i, G, spp)
            part = gpu.render(w, h, n, 50, 0x5EED, capi.FLAG_ACCEL_AUTO, spp_begin=b, want=("sum",))["sum"]
            expect = part if expect is None else expect + part                     # float32, rank order
        H.assert_bits(r["sum"], expect, f"{G} shards, predicted sum")
        H.assert_bits(again["sum"], r["sum"], "run to run")
        assert r["stats"].rays == one["stats"].rays and r["stats"].samples == one["stats"].samples
        assert np.allclose(r["sum"], one["sum"], rtol=2e-6, atol=1e-6)              # only the association of the per-pixel sum differs
        assert (np.abs(r["rgba8"].astype(int) - one["rgba8"].astype(int)) <= 1).all()


@pytest.mark.gpu
def test_one_spp_and_empty_ranges(gpu):
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    with capi.MultiContext([0, 0, 0]) as m:
        m.upload_scene(s); m.set_camera(cam)
        a = m.render(64, 64, 5, 50, 1, capi.FLAG_ACCEL_AUTO | capi.FLAG_ONE_SPP)
        b = gpu.render(64, 64, 5, 50, 1, capi.FLAG_ACCEL_AUTO | capi.FLAG_ONE_SPP)
        assert np.array_equal(a["rgba8"], b["rgba8"]) and [a["info"].spp_count[i] for i in range(3)] == [1, 0, 0]
        a = m.render(64, 64, 2, 50, 1, capi.FLAG_ACCEL_AUTO)                        # fewer samples than shards: one range is empty
        b = gpu.render(64, 64, 2, 50, 1, capi.FLAG_ACCEL_AUTO)
        assert sorted(a["info"].spp_count[i] for i in range(3)) == [0, 1, 1] and np.array_equal(a["rgba8"], b["rgba8"])
        with pytest.raises(capi.MirtError, match="distinct"):
            m.set_option("reduce", capi.REDUCE_RCCL); m.render(8, 8, 2)


@pytest.mark.gpu
def test_rccl_reduce_code_path_with_one_rank(gpu):
    """ncclCommInitAll + grouped ncclReduce through the dlopen'ed librccl, on a one-device communicator (all a one-GPU box has)."""
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    with capi.MultiContext([0]) as m:
        m.set_option("reduce", capi.REDUCE_RCCL)
        m.upload_scene(s); m.set_camera(cam)
        a = m.render(64, 48, 6, 50, 9, capi.FLAG_ACCEL_AUTO)
    b = gpu.render(64, 48, 6, 50, 9, capi.FLAG_ACCEL_AUTO)
    assert a["info"].reduce_kind == capi.REDUCE_RCCL
    H.assert_bits(a["sum"], b["sum"], "one-rank RCCL reduce")


@pytest.mark.gpu
@pytest.mark.skipif(capi.load().mirt_device_count() < 2, reason="needs two GPUs")
def test_two_real_gpus_rccl_and_peer(gpu):
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    one = gpu.render(128, 128, 16, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
    n = min(capi.load().mirt_device_count(), 8)
    for kind in (capi.REDUCE_AUTO, capi.REDUCE_PEER):
        with capi.MultiContext(list(range(n))) as m:
            m.set_option("reduce", kind)
            m.upload_scene(s); m.set_camera(cam)
            r = m.render(128, 128, 16, 50, 0x5EED, capi.FLAG_ACCEL_AUTO)
        assert r["info"].reduce_kind == (capi.REDUCE_PEER if kind == capi.REDUCE_PEER else capi.REDUCE_RCCL)
        assert r["stats"].rays == one["stats"].rays
        assert np.allclose(r["sum"], one["sum"], rtol=2e-6, atol=1e-6)


@pytest.mark.gpu
def test_cpp_tracer_over_several_devices():
    from tests import hostapi
    one = hostapi.render(hostapi.WORLD_CORNELL, 64, 64, 8)
    img, info, rays = hostapi.render_devices(hostapi.WORLD_CORNELL, 64, 64, 8, [0, 0])
    assert info.n_devices == 2 and [info.spp_count[i] for i in range(2)] == [4, 4] and rays > 64 * 64 * 8
    assert (np.abs(img.astype(int) - one.astype(int)) <= 1).all() and (img != one).mean() < 0.01


@pytest.mark.gpu
def test_c5_eight_shards_of_a_4096_frame(gpu):
    """BASELINE config 5 in shape: a 4096^2 frame whose samples are split over EIGHT ranks and summed onto rank 0 (201 MB per
    rank).  One GPU plays all eight (peer reduce, device 0 listed eight times); the sum must be the eight single-range renders
    added in rank order, bit for bit, and equal the one-device image up to re-association."""
    s, cam = scene.cornell_box(1.0)
    gpu.upload_scene(s); gpu.set_camera(cam)
    w = h = 4096; spp = 16
    with capi.MultiContext([0] * 8) as m:
        m.upload_scene(s); m.set_camera(cam)
        r = m.render(w, h, spp, 50, 0xC5, capi.FLAG_ACCEL_AUTO, want=("sum", "rgba8"))
    info = r["info"]
    assert info.n_devices == 8 and info.reduce_bytes == w * h * 12 and [info.spp_count[i] for i in range(8)] == [2] * 8
    expect = None
    for i in range(8):
        part = gpu.render(w, h, 2, 50, 0xC5, capi.FLAG_ACCEL_AUTO, spp_begin=2 * i, want=("sum",))["sum"]
        expect = part if expect is None else expect + part
    assert np.array_equal(r["sum"].view(np.uint32), expect.view(np.uint32))
    one = gpu.render(w, h, spp, 50, 0xC5, capi.FLAG_ACCEL_AUTO, want=("sum", "rgba8"))
    assert r["stats"].rays == one["stats"].rays
    assert np.allclose(r["sum"], one["sum"], rtol=4e-6, atol=2e-6)
    assert (np.abs(r["rgba8"].astype(np.int16) - one["rgba8"].astype(np.int16)) <= 1).all()
