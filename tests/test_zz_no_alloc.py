"""mirt_render_device_noalloc (the entry point that allocates nothing) against mirt_render_device, through the C ABI:
bit-identical radiance sums on two calls in a row, the second with a smaller input, then a BVH scene and the limits."""
import ctypes as C

import numpy as np
import pytest

from raytracinginaweekend_amd import capi, scene

pytestmark = pytest.mark.gpu


class Hip:
    """Device buffers for the test through the HIP runtime libmirt.so itself is linked to (no torch: its bundled
    second HIP runtime does not always find the GPU once this one is up)."""

    def __init__(self):
        capi.load()
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes(self):
        assert self.rt.hipDeviceSynchronize() == 0
        f, t = C.c_size_t(), C.c_size_t()
        assert self.rt.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def zeros(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), nbytes) == 0 and self.rt.hipMemset(p, 0, nbytes) == 0
        return p.value

    def fetch(self, ptr, shape):
        out = np.empty(shape, np.float32)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes, 2) == 0          # hipMemcpyDeviceToHost
        assert self.rt.hipFree(ptr) == 0
        return out


def _both(ctx, hip, w, h, spp, flags, spp_begin=0, depth=50, seed=0x5EED):
    p = capi.RenderParams(w, h, spp_begin, spp, depth, flags, seed)
    a, b = hip.zeros(w * h * 12), hip.zeros(w * h * 12)
    ctx.render_device(p, a, None, capi.Stats())
    ctx.render_device_noalloc(p, b, None)
    return hip.fetch(a, (h, w, 3)), hip.fetch(b, (h, w, 3))


def test_noalloc_matches_render_device():
    hip = Hip()
    with capi.Context(0) as ctx:
        sc, cam = scene.cornell_box(1.0)
        ctx.upload_scene(sc); ctx.set_camera(cam)
        # first call: 3 Mi paths = two batches of the fixed scratch (2 Mi paths each)
        a, b = _both(ctx, hip, 512, 384, 16, capi.FLAG_ACCEL_AUTO)
        assert a.any() and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # second call, smaller input: the scratch still holds the first call's slots and counters
        a, b = _both(ctx, hip, 96, 64, 3, capi.FLAG_ACCEL_AUTO, spp_begin=5)
        assert a.any() and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # 1-SPP and normal display overwrite / end at the first vertex
        for fl in (capi.FLAG_ONE_SPP, capi.FLAG_NORMAL_DISPLAY):
            a, b = _both(ctx, hip, 64, 64, 2, capi.FLAG_ACCEL_AUTO | fl)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # a BVH scene: the traversal-stack spill comes out of the scratch as well
        sc, cam = scene.spheres_world(1.5)
        ctx.upload_scene(sc); ctx.set_camera(cam)
        for fl in (capi.FLAG_ACCEL_AUTO, 0):
            a, b = _both(ctx, hip, 120, 80, 4, fl)
            assert a.any() and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # ... and no device memory is taken by a call: free bytes before and after a (warmed) call are the same, on the list walk, on
        # the BVH walks (whose kernels use a few dwords of private scratch per lane: the runtime sizes that at their first launch,
        # which the calls above were) and on an image of more than 2 Mi pixels, i.e. several pixel tiles of the fixed scratch
        big_w, big_h = 2048, 1200                               # 2.46 Mi pixels > the 2 Mi paths of one batch
        outs = {k: hip.zeros(n) for k, n in (("small", 120 * 80 * 12), ("big", big_w * big_h * 12))}
        ref_big = hip.zeros(big_w * big_h * 12)
        for fl in (capi.FLAG_ACCEL_AUTO, 0):
            ctx.render_device_noalloc(capi.RenderParams(big_w, big_h, 0, 1, 50, fl, 9), outs["big"], None)     # warm: first launch of this size
        ctx.render_device(capi.RenderParams(big_w, big_h, 0, 2, 50, capi.FLAG_ACCEL_AUTO, 9), ref_big, None, capi.Stats())
        assert hip.rt.hipMemset(outs["big"], 0, big_w * big_h * 12) == 0
        before = hip.free_bytes()
        ctx.render_device_noalloc(capi.RenderParams(120, 80, 0, 4, 50, capi.FLAG_ACCEL_AUTO, 9), outs["small"], None)
        ctx.render_device_noalloc(capi.RenderParams(120, 80, 4, 4, 50, 0, 9), outs["small"], None)
        ctx.render_device_noalloc(capi.RenderParams(big_w, big_h, 0, 2, 50, capi.FLAG_ACCEL_AUTO, 9), outs["big"], None)
        assert hip.free_bytes() == before, "mirt_render_device_noalloc took device memory"
        a, b = hip.fetch(ref_big, (big_h, big_w, 3)), hip.fetch(outs["big"], (big_h, big_w, 3))
        assert a.any() and np.array_equal(a.view(np.uint32), b.view(np.uint32))       # two pixel tiles, same bits
        hip.fetch(outs["small"], (80, 120, 3))
        # beyond the fixed scratch: a clear error, not an allocation
        p = capi.RenderParams(4096, 4096, 0, 1024, 50, capi.FLAG_ACCEL_AUTO, 1)
        d = hip.zeros(64)
        with pytest.raises(capi.MirtError, match="batches"):
            ctx.render_device_noalloc(p, d, None)
        ctx.set_option("engine", 0)
        with pytest.raises(capi.MirtError, match="persistent"):
            ctx.render_device_noalloc(capi.RenderParams(8, 8, 0, 1, 50, 0, 1), d, None)
