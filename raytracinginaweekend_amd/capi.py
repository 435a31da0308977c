"""ctypes binding of libmirt.so (include/mirt.h), the C ABI of the gfx950 path tracer.

There is no CPU fallback: importing works anywhere, but `load()` raises if the HIP library has not
been built and `Context()` raises if no GPU is present.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmirt.so")

PRIM_SPHERE, PRIM_RECT = 0, 1
MAT_DIFFUSE_LIGHT, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2, 3
CHILD_NONE = -(1 << 31)
OP_TRANSLATE, OP_ROTATE_X, OP_ROTATE_Y, OP_ROTATE_Z = 0, 1, 2, 3
MAX_CHAIN_OPS = 8
FLAG_ONE_SPP, FLAG_NORMAL_DISPLAY, FLAG_ACCEL_LIST, FLAG_ACCEL_AUTO, FLAG_ACCEL_ORDERED = 1, 2, 4, 16, 32

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def axes_pack(a, b, c, rev, rot):
    return a | (b << 2) | (c << 4) | (int(bool(rev)) << 6) | (int(bool(rot)) << 7)


class SceneStruct(C.Structure):
    """mirt_scene (and, layout-identical, the oracle's orc_scene)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_prims", C.c_uint32), ("prim_kind", u32p), ("prim_off", f32p), ("prim_geom", f32p),
        ("prim_axes", u32p), ("prim_rot", f32p), ("prim_mat", u32p),
        ("n_objects", C.c_uint32), ("obj_first", u32p), ("obj_count", u32p),
        ("n_mats", C.c_uint32), ("mat_kind", u32p), ("mat_albedo", f32p), ("mat_fuzz", f32p),
        ("mat_ior", f32p), ("mat_emit", f32p),
        ("n_nodes", C.c_uint32), ("node_min", f32p), ("node_max", f32p), ("node_left", i32p),
        ("node_right", i32p), ("root", C.c_int32), ("ambient", C.c_float * 3),
        ("n_textures", C.c_uint32), ("tex_width", u32p), ("tex_height", u32p), ("tex_offset", u32p),
        ("tex_rgba8", u8p), ("mat_texture", i32p),
        ("prim_chain", u32p), ("prim_nops", u32p), ("prim_center", f32p), ("n_ops", C.c_uint32), ("op_kind", u32p), ("op_arg", f32p),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(SceneStruct)


class CameraStruct(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("top_left", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("lens_radius", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("spp_begin", C.c_uint32),
                ("spp_count", C.c_uint32), ("max_depth", C.c_uint32), ("flags", C.c_uint32),
                ("seed", C.c_uint64)]

    def __init__(self, width=0, height=0, spp_begin=0, spp_count=0, max_depth=50, flags=0, seed=0):
        super().__init__(C.sizeof(RenderParams), width, height, spp_begin, spp_count, max_depth, flags, seed)


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("rays", C.c_uint64), ("samples", C.c_uint64), ("depth_hist", C.c_uint64 * 64),
                ("device_ms", C.c_double), ("extend_ms", C.c_double), ("bounce_launches", C.c_uint32),
                ("batches", C.c_uint32), ("first_ms", C.c_double), ("extend_launches", C.c_uint32),
                ("first_launches", C.c_uint32), ("depth_ms", C.c_double * 64)]

    def __init__(self):
        super().__init__()
        self.struct_size = C.sizeof(Stats)


ABI_VERSION = 3
MULTI_MAX_DEVICES = 16
REDUCE_AUTO, REDUCE_PEER, REDUCE_RCCL = 0, 1, 2


class MultiInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_devices", C.c_uint32), ("reduce_kind", C.c_uint32), ("pad", C.c_uint32),
                ("render_ms", C.c_double), ("reduce_ms", C.c_double), ("resolve_ms", C.c_double), ("reduce_bytes", C.c_uint64),
                ("device", C.c_int32 * MULTI_MAX_DEVICES), ("spp_begin", C.c_uint32 * MULTI_MAX_DEVICES),
                ("spp_count", C.c_uint32 * MULTI_MAX_DEVICES), ("device_ms", C.c_double * MULTI_MAX_DEVICES)]

    def __init__(self):
        super().__init__()
        self.struct_size = C.sizeof(MultiInfo)


HIT_DTYPE = np.dtype([("t", "<f4"), ("pos", "<f4", 3), ("nrm", "<f4", 3), ("u", "<f4"), ("v", "<f4"),
                      ("mat", "<i4"), ("prim", "<i4")])

EXPORTS = [
    "mirt_abi_version", "mirt_check_abi",
    "mirt_create", "mirt_destroy", "mirt_last_error", "mirt_device_count", "mirt_runtime_version", "mirt_set_option",
    "mirt_upload_scene", "mirt_build_bvh", "mirt_set_camera", "mirt_render_device", "mirt_render_device_noalloc", "mirt_resolve_device", "mirt_render", "mirt_render_progressive",
    "mirt_aabb_hit_batch", "mirt_prim_hit_batch", "mirt_world_hit_batch", "mirt_scatter_batch",
    "mirt_multi_create", "mirt_multi_destroy", "mirt_multi_last_error", "mirt_multi_device_count", "mirt_multi_set_option",
    "mirt_multi_upload_scene", "mirt_multi_set_camera", "mirt_multi_spp_range", "mirt_multi_render",
    "mirt_camera_rays", "mirt_philox_batch", "mirt_arith_batch", "mirt_tex_sample_batch", "mirt_write_ppm", "mirt_write_image",
]

_lib = None


class MirtError(RuntimeError):
    pass


def load():
    """dlopen libmirt.so; raises MirtError when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MIRT_LIB", LIB_PATH)       # MIRT_LIB: a diagnostic build of the same library (tools/)
    if not os.path.exists(path):
        raise MirtError(f"{path} is missing: build it with `python -m raytracinginaweekend_amd.build` "
                        "(the path tracer has no CPU fallback)")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.mirt_abi_version.argtypes = []
    lib.mirt_check_abi.argtypes = [C.c_uint32] + [C.c_size_t] * 5
    lib.mirt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mirt_destroy.argtypes = [vp]; lib.mirt_destroy.restype = None
    lib.mirt_last_error.argtypes = [vp]; lib.mirt_last_error.restype = C.c_char_p
    lib.mirt_device_count.argtypes = []
    lib.mirt_runtime_version.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mirt_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.mirt_upload_scene.argtypes = [vp, C.POINTER(SceneStruct)]
    lib.mirt_set_camera.argtypes = [vp, C.POINTER(CameraStruct)]
    lib.mirt_build_bvh.argtypes = [C.POINTER(SceneStruct), C.c_uint32, vp, vp, vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mirt_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    lib.mirt_render_device_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    lib.mirt_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.mirt_multi_destroy.argtypes = [vp]; lib.mirt_multi_destroy.restype = None
    lib.mirt_multi_last_error.argtypes = [vp]; lib.mirt_multi_last_error.restype = C.c_char_p
    lib.mirt_multi_device_count.argtypes = [vp]
    lib.mirt_multi_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.mirt_multi_upload_scene.argtypes = [vp, C.POINTER(SceneStruct)]
    lib.mirt_multi_set_camera.argtypes = [vp, C.POINTER(CameraStruct)]
    lib.mirt_multi_spp_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]; lib.mirt_multi_spp_range.restype = None
    lib.mirt_multi_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, C.POINTER(Stats), C.POINTER(MultiInfo)]
    lib.mirt_resolve_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    lib.mirt_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, C.POINTER(Stats)]
    lib.mirt_render_progressive.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, C.POINTER(Stats)]
    lib.mirt_aabb_hit_batch.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t, C.c_float, C.c_float, vp]
    lib.mirt_prim_hit_batch.argtypes = [vp, C.c_uint32, vp, vp, C.c_size_t, C.c_float, C.c_float, vp]
    lib.mirt_world_hit_batch.argtypes = [vp, vp, vp, C.c_size_t, C.c_float, C.c_float, C.c_uint32, vp]
    lib.mirt_scatter_batch.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.mirt_camera_rays.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, vp, vp, vp]
    lib.mirt_philox_batch.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.mirt_arith_batch.argtypes = [vp, C.c_uint32, vp, vp, C.c_size_t, vp]
    lib.mirt_tex_sample_batch.argtypes = [vp, C.c_uint32, vp, C.c_size_t, vp]
    lib.mirt_write_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    lib.mirt_write_image.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, vp, vp]
    for name in EXPORTS:
        if name not in ("mirt_destroy", "mirt_last_error", "mirt_abi_version", "mirt_multi_destroy", "mirt_multi_last_error", "mirt_multi_spp_range"):
            getattr(lib, name).restype = C.c_int
    lib.mirt_abi_version.restype = C.c_uint32
    # the ABI guard of include/mirt.h: this mirror's struct sizes against the library's, once at load
    if lib.mirt_check_abi(ABI_VERSION, C.sizeof(SceneStruct), C.sizeof(CameraStruct), HIT_DTYPE.itemsize, C.sizeof(RenderParams), C.sizeof(Stats)) != 0:
        raise MirtError("capi.py and libmirt.so disagree about the ABI: " + lib.mirt_last_error(None).decode())
    _lib = lib
    check_single_hip_runtime()
    return lib


def mapped_hip_runtimes(maps_text=None):
    """The distinct libamdhip64 files mapped into this process (real paths), from /proc/self/maps."""
    if maps_text is None:
        with open("/proc/self/maps") as f:
            maps_text = f.read()
    seen = []
    for line in maps_text.splitlines():
        parts = line.split(None, 5)
        if len(parts) == 6 and os.path.basename(parts[5]).startswith("libamdhip64.so"):
            p = os.path.realpath(parts[5].removesuffix(" (deleted)"))
            if p not in seen:
                seen.append(p)
    return seen


def check_single_hip_runtime(maps_text=None):
    """libmirt.so is linked against /opt/rocm's libamdhip64.so.7; torch bundles a libamdhip64.so of its own.  Device pointers
    and stream handles cross between the two users (bench.py hands torch's to libmirt), which is sound only when both
    resolved to ONE mapped runtime: the loader shares a library by SONAME when torch is imported FIRST (or not at all);
    with libmirt loaded first, torch's loader finds its own copy by path and the process holds two runtimes, each with its
    own device state.  Raises MirtError in that case; returns the one mapped path (None: none mapped yet)."""
    rts = mapped_hip_runtimes(maps_text)
    if len(rts) > 1:
        raise MirtError("two HIP runtimes are mapped into this process: " + ", ".join(rts) + " -- import torch BEFORE "
                        "raytracinginaweekend_amd.capi.load() (the loader then shares torch's libamdhip64 by SONAME), or do not import torch")
    return rts[0] if rts else None


def hip_runtime_info():
    """{"path": the one libamdhip64 mapped, "hip_runtime": hipRuntimeGetVersion, "hip_driver": hipDriverGetVersion} as libmirt sees them."""
    lib = load()
    rt, drv = C.c_int(), C.c_int()
    if lib.mirt_runtime_version(C.byref(rt), C.byref(drv)) != 0:
        raise MirtError("mirt_runtime_version failed: " + lib.mirt_last_error(None).decode())
    return {"path": check_single_hip_runtime(), "hip_runtime": rt.value, "hip_driver": drv.value}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


class Context:
    """One mirt_ctx bound to one GPU."""

    def __init__(self, device=0):
        self.lib = load()
        check_single_hip_runtime()                     # torch imported after load() would have brought a second one
        h = C.c_void_p()
        rc = self.lib.mirt_create(int(device), C.byref(h))
        if rc != 0:
            raise MirtError(f"mirt_create failed ({rc}): {self.lib.mirt_last_error(None).decode()}")
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.mirt_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc, what):
        if rc != 0:
            raise MirtError(f"{what} failed ({rc}): {self.lib.mirt_last_error(self.h).decode()}")

    def set_option(self, name, value):
        self._ck(self.lib.mirt_set_option(self.h, name.encode(), int(value)), "mirt_set_option")

    def upload_scene(self, scene):
        """scene: raytracinginaweekend_amd.scene.FlatScene"""
        st = scene.as_struct()
        self._ck(self.lib.mirt_upload_scene(self.h, C.byref(st)), "mirt_upload_scene")

    def set_camera(self, cam):
        self._ck(self.lib.mirt_set_camera(self.h, C.byref(cam)), "mirt_set_camera")

    def render(self, width, height, spp, max_depth=50, seed=0x5EED, flags=FLAG_ACCEL_AUTO, spp_begin=0,
               want=("sum", "pixels", "rgba8")):
        p = RenderParams(width, height, spp_begin, spp, max_depth, flags, seed)
        n = width * height
        out = {}
        s = np.empty((height, width, 3), np.float32) if "sum" in want else None
        px = np.empty((height, width, 3), np.float32) if "pixels" in want else None
        rg = np.empty((height, width, 4), np.uint8) if "rgba8" in want else None
        st = Stats()
        self._ck(self.lib.mirt_render(self.h, C.byref(p), _ptr(s) if s is not None else None,
                                      _ptr(px) if px is not None else None,
                                      _ptr(rg) if rg is not None else None, C.byref(st)), "mirt_render")
        out["sum"], out["pixels"], out["rgba8"], out["stats"] = s, px, rg, st
        return out

    def render_progressive(self, width, height, spp_begin, spp_count, max_depth=50, seed=0x5EED, flags=FLAG_ACCEL_AUTO,
                           want=("sum", "pixels", "rgba8")):
        """mirt_render_progressive: adds samples [spp_begin, spp_begin+spp_count) to the image kept on the device."""
        p = RenderParams(width, height, spp_begin, spp_count, max_depth, flags, seed)
        s = np.empty((height, width, 3), np.float32) if "sum" in want else None
        px = np.empty((height, width, 3), np.float32) if "pixels" in want else None
        rg = np.empty((height, width, 4), np.uint8) if "rgba8" in want else None
        st = Stats()
        self._ck(self.lib.mirt_render_progressive(self.h, C.byref(p), _ptr(s) if s is not None else None,
                                                  _ptr(px) if px is not None else None,
                                                  _ptr(rg) if rg is not None else None, C.byref(st)), "mirt_render_progressive")
        return {"sum": s, "pixels": px, "rgba8": rg, "stats": st}

    def render_device(self, params, d_sum_ptr, stream=None, stats=None):
        self._ck(self.lib.mirt_render_device(self.h, C.byref(params), C.c_void_p(d_sum_ptr),
                                             C.c_void_p(stream) if stream else None,
                                             C.byref(stats) if stats is not None else None), "mirt_render_device")

    def render_device_noalloc(self, params, d_sum_ptr, stream=None):
        """mirt_render_device_noalloc: same image, scratch in the library's fixed device array, no allocation."""
        self._ck(self.lib.mirt_render_device_noalloc(self.h, C.byref(params), C.c_void_p(d_sum_ptr),
                                                     C.c_void_p(stream) if stream else None), "mirt_render_device_noalloc")

    def resolve_device(self, d_sum_ptr, width, height, spp_total, flags, d_pixels_ptr=None, d_rgba8_ptr=None, stream=None):
        self._ck(self.lib.mirt_resolve_device(self.h, C.c_void_p(d_sum_ptr), width, height, spp_total, flags,
                                              C.c_void_p(d_pixels_ptr) if d_pixels_ptr else None,
                                              C.c_void_p(d_rgba8_ptr) if d_rgba8_ptr else None,
                                              C.c_void_p(stream) if stream else None), "mirt_resolve_device")

    # ---- function-level entry points -------------------------------------------------------
    def tex_sample_batch(self, tex, uv):
        uv = _f32(uv, (-1, 2))
        out = np.zeros((uv.shape[0], 3), np.float32)
        self._ck(self.lib.mirt_tex_sample_batch(self.h, tex, _ptr(uv), uv.shape[0], _ptr(out)), "mirt_tex_sample_batch")
        return out

    def aabb_hit_batch(self, bmin, bmax, org, dir_, t_min, t_max):
        bmin, bmax, org, dir_ = _f32(bmin, (-1, 3)), _f32(bmax, (-1, 3)), _f32(org, (-1, 3)), _f32(dir_, (-1, 3))
        out = np.zeros((org.shape[0], bmin.shape[0]), np.uint8)
        self._ck(self.lib.mirt_aabb_hit_batch(self.h, _ptr(bmin), _ptr(bmax), bmin.shape[0], _ptr(org), _ptr(dir_),
                                              org.shape[0], t_min, t_max, _ptr(out)), "mirt_aabb_hit_batch")
        return out

    def prim_hit_batch(self, prim, org, dir_, t_min, t_max):
        org, dir_ = _f32(org, (-1, 3)), _f32(dir_, (-1, 3))
        out = np.zeros(org.shape[0], HIT_DTYPE)
        self._ck(self.lib.mirt_prim_hit_batch(self.h, prim, _ptr(org), _ptr(dir_), org.shape[0], t_min, t_max, _ptr(out)),
                 "mirt_prim_hit_batch")
        return out

    def world_hit_batch(self, org, dir_, t_min, t_max, flags=0):
        org, dir_ = _f32(org, (-1, 3)), _f32(dir_, (-1, 3))
        out = np.zeros(org.shape[0], HIT_DTYPE)
        self._ck(self.lib.mirt_world_hit_batch(self.h, _ptr(org), _ptr(dir_), org.shape[0], t_min, t_max, flags, _ptr(out)),
                 "mirt_world_hit_batch")
        return out

    def scatter_batch(self, hits, dir_in, draws):
        hits = np.ascontiguousarray(hits, HIT_DTYPE); dir_in = _f32(dir_in, (-1, 3))
        draws = np.ascontiguousarray(draws, np.uint32); n = hits.shape[0]
        alive = np.zeros(n, np.uint8); att = np.zeros((n, 3), np.float32); org = np.zeros((n, 3), np.float32)
        dr = np.zeros((n, 3), np.float32); em = np.zeros((n, 3), np.float32); used = np.zeros(n, np.uint32)
        self._ck(self.lib.mirt_scatter_batch(self.h, _ptr(hits), _ptr(dir_in), n, _ptr(draws), draws.shape[1], _ptr(alive),
                                             _ptr(att), _ptr(org), _ptr(dr), _ptr(em), _ptr(used)), "mirt_scatter_batch")
        return dict(alive=alive, atten=att, org=org, dir=dr, emit=em, consumed=used)

    def camera_rays(self, uv, draws):
        uv = _f32(uv, (-1, 2)); draws = np.ascontiguousarray(draws, np.uint32); n = uv.shape[0]
        org = np.zeros((n, 3), np.float32); dr = np.zeros((n, 3), np.float32); used = np.zeros(n, np.uint32)
        self._ck(self.lib.mirt_camera_rays(self.h, _ptr(uv), n, _ptr(draws), draws.shape[1], _ptr(org), _ptr(dr), _ptr(used)),
                 "mirt_camera_rays")
        return dict(org=org, dir=dr, consumed=used)

    def philox_batch(self, ctr, key):
        ctr = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 4); key = np.ascontiguousarray(key, np.uint32)
        out = np.zeros_like(ctr)
        self._ck(self.lib.mirt_philox_batch(self.h, _ptr(ctr), ctr.shape[0], _ptr(key), _ptr(out)), "mirt_philox_batch")
        return out

    def arith_batch(self, op, x, y=None):
        x = np.ascontiguousarray(x).view(np.float32).ravel()
        yy = None if y is None else np.ascontiguousarray(y, np.float32).ravel()
        out = np.zeros_like(x)
        self._ck(self.lib.mirt_arith_batch(self.h, op, _ptr(x), _ptr(yy) if yy is not None else None, x.shape[0], _ptr(out)),
                 "mirt_arith_batch")
        return out


class MultiContext:
    """mirt_multi: one render sharded by samples per pixel over several GPUs of this process (a device may be listed
    twice with the peer reduce, which is how a one-GPU box exercises the path)."""

    def __init__(self, devices):
        self.lib = load()
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.mirt_multi_create(ids, len(devices), C.byref(h))
        if rc != 0:
            raise MirtError(f"mirt_multi_create failed ({rc}): {self.lib.mirt_multi_last_error(None).decode()}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.mirt_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc, what):
        if rc != 0:
            raise MirtError(f"{what} failed ({rc}): {self.lib.mirt_multi_last_error(self.h).decode()}")

    def set_option(self, name, value):
        self._ck(self.lib.mirt_multi_set_option(self.h, name.encode(), int(value)), "mirt_multi_set_option")

    def upload_scene(self, scene):
        st = scene.as_struct()
        self._ck(self.lib.mirt_multi_upload_scene(self.h, C.byref(st)), "mirt_multi_upload_scene")

    def set_camera(self, cam):
        self._ck(self.lib.mirt_multi_set_camera(self.h, C.byref(cam)), "mirt_multi_set_camera")

    def render(self, width, height, spp, max_depth=50, seed=0x5EED, flags=FLAG_ACCEL_AUTO, spp_begin=0, want=("sum", "pixels", "rgba8")):
        p = RenderParams(width, height, spp_begin, spp, max_depth, flags, seed)
        s = np.empty((height, width, 3), np.float32) if "sum" in want else None
        px = np.empty((height, width, 3), np.float32) if "pixels" in want else None
        rg = np.empty((height, width, 4), np.uint8) if "rgba8" in want else None
        st, info = Stats(), MultiInfo()
        self._ck(self.lib.mirt_multi_render(self.h, C.byref(p), _ptr(s) if s is not None else None, _ptr(px) if px is not None else None,
                                            _ptr(rg) if rg is not None else None, C.byref(st), C.byref(info)), "mirt_multi_render")
        return {"sum": s, "pixels": px, "rgba8": rg, "stats": st, "info": info}


def multi_spp_range(rank, world, spp_begin, spp_count):
    b, n = C.c_uint32(), C.c_uint32()
    load().mirt_multi_spp_range(rank, world, spp_begin, spp_count, C.byref(b), C.byref(n))
    return b.value, n.value


def build_bvh(scene):
    """The library's own (host-side, binned SAH) BVH for `scene`; attaches it to the scene and returns its depth."""
    st = scene.as_struct()
    cap = 2 * scene.n_objects + 2
    mn = np.zeros((cap, 3), np.float32); mx = np.zeros((cap, 3), np.float32)
    l = np.zeros(cap, np.int32); r = np.zeros(cap, np.int32)
    root = C.c_int32(); n = C.c_uint32(); depth = C.c_uint32()
    rc = load().mirt_build_bvh(C.byref(st), cap, _ptr(mn), _ptr(mx), _ptr(l), _ptr(r), C.byref(root), C.byref(n), C.byref(depth))
    if rc != 0:
        raise MirtError(f"mirt_build_bvh failed ({rc})")
    scene.set_bvh(mn[:n.value], mx[:n.value], l[:n.value], r[:n.value], root.value)
    return depth.value


IMG_P3, IMG_P6, IMG_PFM = 0, 1, 2


def write_image(path, fmt, rgba8=None, rgb_float=None):
    """mirt_write_image: ASCII P3 / binary P6 from RGBA8, PFM from float RGB (row 0 = top on input)."""
    a = np.ascontiguousarray(rgba8, np.uint8) if rgba8 is not None else None
    f = np.ascontiguousarray(rgb_float, np.float32) if rgb_float is not None else None
    h, w = (a if a is not None else f).shape[:2]
    rc = load().mirt_write_image(os.fsencode(path), fmt, w, h, _ptr(a) if a is not None else None, _ptr(f) if f is not None else None)
    if rc != 0:
        raise MirtError(f"mirt_write_image failed ({rc})")


def write_ppm(path, rgba8):
    """PPMImageMaker::OutputRGBA8ToFile (ASCII P3)."""
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    h, w = rgba8.shape[:2]
    rc = load().mirt_write_ppm(os.fsencode(path), w, h, _ptr(rgba8))
    if rc != 0:
        raise MirtError(f"mirt_write_ppm failed ({rc})")
