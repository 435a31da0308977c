"""ctypes access to libmirt_host.so's harness shim (the drop-in C++ scene API)."""
import ctypes as C
import os

import numpy as np

from raytracinginaweekend_amd import capi, scene

LIB = os.path.join(os.path.dirname(capi.LIB_PATH), "libmirt_host.so")
WORLD_RANDOM_SPHERES, WORLD_CORNELL, WORLD_FOUR_SPHERES, WORLD_TWO_BOXES, WORLD_TEXTURED, WORLD_CHAINS = 0, 1, 2, 3, 4, 5
_lib = None


def ensure_built():
    """The host layer is plain g++ (no hipcc): rebuild it in-tree if a fresh checkout lacks the binaries."""
    libdir = os.path.dirname(capi.LIB_PATH)
    if not all(os.path.exists(os.path.join(libdir, n)) for n in ("libmirt_host.so", "cornell_example", "flythrough")):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(libdir), "host")], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        capi.load()
        ensure_built()
        L = C.CDLL(LIB)
        L.mirt_host_world_create.argtypes = [C.c_int, C.c_float, C.c_uint64, C.c_char_p, C.c_size_t]; L.mirt_host_world_create.restype = C.c_void_p
        L.mirt_host_world_scene.argtypes = [C.c_void_p]; L.mirt_host_world_scene.restype = C.POINTER(capi.SceneStruct)
        L.mirt_host_world_camera.argtypes = [C.c_void_p]; L.mirt_host_world_camera.restype = C.POINTER(capi.CameraStruct)
        L.mirt_host_world_destroy.argtypes = [C.c_void_p]; L.mirt_host_world_destroy.restype = None
        L.mirt_host_render.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                       C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.mirt_host_render.restype = C.c_int
        L.mirt_host_try_chain.argtypes = [C.c_uint32, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
        L.mirt_host_try_chain.restype = C.c_int
        _lib = L
    return _lib


def _arr(ptr, n, k=1):
    a = np.ctypeslib.as_array(ptr, shape=(n * k,)).copy()
    return a.reshape(n, k) if k > 1 else a


def build_world(world_id, aspect, seed):
    """World::ConstructWorld through the C++ API -> (FlatScene with the host's tree attached, camera)."""
    err = C.create_string_buffer(512)
    h = lib().mirt_host_world_create(world_id, aspect, seed, err, 512)
    if not h:
        raise RuntimeError(err.value.decode())
    try:
        s = lib().mirt_host_world_scene(h).contents
        fs = scene.FlatScene(ambient=tuple(s.ambient))
        n, no, nm, nn = s.n_prims, s.n_objects, s.n_mats, s.n_nodes
        fs.kind = _arr(s.prim_kind, n).tolist(); fs.off = _arr(s.prim_off, n, 3); fs.geom = _arr(s.prim_geom, n, 5)
        fs.axes = _arr(s.prim_axes, n).tolist(); fs.rot = _arr(s.prim_rot, n, 6); fs.pmat = _arr(s.prim_mat, n).tolist()
        fs.obj_first = _arr(s.obj_first, no).tolist(); fs.obj_count = _arr(s.obj_count, no).tolist()
        fs.chain = [None] * n; fs.center = [(0.0, 0.0, 0.0)] * n
        if s.prim_chain:                              # chain-form primitives: their ops, outermost first, and the spheres' own centres
            first, nops, ctr = _arr(s.prim_chain, n), _arr(s.prim_nops, n), _arr(s.prim_center, n, 3)
            kinds = _arr(s.op_kind, max(1, s.n_ops)); args = _arr(s.op_arg, max(1, s.n_ops), 3)
            fs.center = [tuple(float(x) for x in c) for c in ctr]
            fs.chain = [None if not first[p] else [(int(kinds[first[p] - 1 + k]), tuple(float(x) for x in args[first[p] - 1 + k])) for k in range(nops[p])] for p in range(n)]
        fs.mkind = _arr(s.mat_kind, nm).tolist(); fs.albedo = _arr(s.mat_albedo, nm, 3); fs.fuzz = _arr(s.mat_fuzz, nm)
        fs.ior = _arr(s.mat_ior, nm); fs.emit = _arr(s.mat_emit, nm, 3)
        if s.n_textures:
            tw, th, to = _arr(s.tex_width, s.n_textures), _arr(s.tex_height, s.n_textures), _arr(s.tex_offset, s.n_textures)
            total = int(to[-1]) + int(tw[-1]) * int(th[-1])
            texels = np.ctypeslib.as_array(s.tex_rgba8, shape=(total * 4,)).copy()
            fs.textures = [texels[4 * int(o): 4 * (int(o) + int(w_) * int(h_))].reshape(int(h_), int(w_), 4) for w_, h_, o in zip(tw, th, to)]
            fs.mtex = _arr(s.mat_texture, nm).tolist()
        else:
            fs.mtex = [-1] * nm
        fs.set_bvh(_arr(s.node_min, nn, 3), _arr(s.node_max, nn, 3), _arr(s.node_left, nn), _arr(s.node_right, nn), s.root)
        cam = capi.CameraStruct.from_buffer_copy(bytes(lib().mirt_host_world_camera(h).contents))
        return fs, cam
    finally:
        lib().mirt_host_world_destroy(h)


def try_chain(n_translates, cx=0.0):
    """Translate^n(Sphere(cx,0,0)) through Object::Flatten -> (n_prims, n_ops, chain form?) or raises with the flattener's message."""
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    err = C.create_string_buffer(512)
    if lib().mirt_host_try_chain(n_translates, cx, C.byref(a), C.byref(b), C.byref(c), err, 512) != 0:
        raise RuntimeError(err.value.decode())
    return a.value, b.value, bool(c.value)


def render(world_id, w, h, spp, depth=50, scene_seed=1, seed=0x5EED, one_spp=False, normal=False, ppm_dir=None):
    out = np.zeros((h, w, 4), np.uint8)
    err = C.create_string_buffer(512)
    rc = lib().mirt_host_render(world_id, w, h, spp, depth, scene_seed, seed, int(one_spp), int(normal), out.ctypes.data_as(C.c_void_p),
                                os.fsencode(ppm_dir) if ppm_dir else None, err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return out


def render_rebuilt(world_a, world_b, w, h, spp, scene_seed=1, seed=0x5EED):
    """One World + one HomemadeRayTracer: world_a rendered, the world deconstructed and rebuilt as world_b, rendered again."""
    L = lib()
    L.mirt_host_render_rebuilt.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.mirt_host_render_rebuilt.restype = C.c_int
    a, b = np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)
    err = C.create_string_buffer(512)
    if L.mirt_host_render_rebuilt(world_a, world_b, w, h, spp, scene_seed, seed, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), err, 512) != 0:
        raise RuntimeError(err.value.decode())
    return a, b


def render_devices(world_id, w, h, spp, devices, scene_seed=1, seed=0x5EED):
    """HomemadeRayTracer::SetDevices(devices) + TraceRay -> (rgba8, MultiInfo, rays)."""
    L = lib()
    L.mirt_host_render_devices.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_int), C.c_int, C.c_void_p,
                                           C.POINTER(capi.MultiInfo), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    L.mirt_host_render_devices.restype = C.c_int
    out = np.zeros((h, w, 4), np.uint8)
    info, rays = capi.MultiInfo(), C.c_uint64()
    err = C.create_string_buffer(512)
    ids = (C.c_int * len(devices))(*devices)
    if L.mirt_host_render_devices(world_id, w, h, spp, scene_seed, seed, ids, len(devices), out.ctypes.data_as(C.c_void_p), C.byref(info), C.byref(rays), err, 512) != 0:
        raise RuntimeError(err.value.decode())
    return out, info, rays.value


def decode_tga(path, cap=1 << 24):
    L = lib()
    L.mirt_host_decode_tga.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.mirt_host_decode_tga.restype = C.c_int
    w, h = C.c_uint32(), C.c_uint32()
    buf = np.zeros(cap, np.uint8)
    err = C.create_string_buffer(512)
    if L.mirt_host_decode_tga(os.fsencode(path), C.byref(w), C.byref(h), buf.ctypes.data_as(C.c_void_p), cap, err, 512) != 0:
        raise RuntimeError(err.value.decode())
    return buf[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()


WORLD_TEXTURED = 4
KEY = {"W": ord("W"), "A": ord("A"), "S": ord("S"), "D": ord("D"), "L": 0x25, "U": 0x26, "R": 0x27, "P": 0x28}


def camera_script(world_id, aspect, scene_seed, frames, dt):
    """frames: list of strings of keys from KEY held during that frame -> one CameraStruct per frame."""
    L = lib()
    L.mirt_host_camera_script.argtypes = [C.c_int, C.c_float, C.c_uint64, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_char_p, C.c_size_t]
    L.mirt_host_camera_script.restype = C.c_int
    keys = np.zeros((len(frames), 4), np.uint8)
    for f, held in enumerate(frames):
        for k, ch in enumerate(held):
            keys[f, k] = KEY[ch]
    out = (capi.CameraStruct * len(frames))()
    err = C.create_string_buffer(512)
    rc = L.mirt_host_camera_script(world_id, aspect, scene_seed, keys.ctypes.data_as(C.c_void_p), len(frames), dt, C.cast(out, C.c_void_p), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return [np.frombuffer(bytes(c), np.float32).copy() for c in out]
