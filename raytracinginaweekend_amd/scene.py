"""Python mirror of the reference's scene-building API for harnesses (tests, bench, smoke).

The drop-in host API is the C++ one under host/; this module builds the same flattened scenes
(include/mirt.h `mirt_scene`) from Python with the reference's factory semantics:
SimpleObjectSphere / SimpleObjectRect / SimpleObjectCube (SimpleObject.cpp:129-184,249-279),
the material table of Resources::LoadMaterials (Resouces.cpp:104-209), the two worlds of
World::ConstructWorld (World.cpp:25-101) and SimpleCamera::Initialize/InternalUpdate
(SimpleCamera.cpp:32-53,139-191).  All arithmetic is float32, in the reference's order.
Citations are relative to /root/reference/RayTracer/RayTracer/.
"""
import math

import numpy as np

from . import capi

F = np.float32
XY_RECT, XZ_RECT, ZY_RECT = 0, 1, 2          # SimpleObject.h:57-62


class FlatScene:
    """Accumulates objects the way the reference's factories do and flattens to SoA arrays."""

    def __init__(self, ambient=(0.0, 0.0, 0.0)):
        self.kind, self.off, self.geom, self.axes, self.rot, self.pmat = [], [], [], [], [], []
        self.chain, self.center = [], []   # per primitive: None (factory form) or its instance ops, outermost first; SphereHitable::m_center
        self.obj_first, self.obj_count = [], []
        self.mkind, self.albedo, self.fuzz, self.ior, self.emit, self.mtex = [], [], [], [], [], []
        self.textures = []    # (h, w, 4) uint8 arrays, row 0 = top (SimpleTexture2D_TGAImage::m_pixelData)
        self.ambient = tuple(float(x) for x in ambient)
        self.nodes = None     # optional (node_min, node_max, left, right, root)
        self._arrays = None

    # ---- materials (Materials.cpp) ---------------------------------------------------------
    def _mat(self, kind, albedo=(0, 0, 0), fuzz=0.0, ior=0.0, emit=(0, 0, 0), texture=-1):
        self.mkind.append(kind); self.albedo.append(albedo); self.fuzz.append(fuzz); self.ior.append(ior); self.emit.append(emit)
        self.mtex.append(int(texture))
        self._arrays = None
        return len(self.mkind) - 1

    def texture(self, rgba8):
        """SimpleTexture2D_TGAImage (SimpleTexture2D.cpp:109-118) from decoded texels, (h, w, 4) uint8, row 0 = top."""
        t = np.ascontiguousarray(rgba8, np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        self.textures.append(t); self._arrays = None
        return len(self.textures) - 1

    def lambertian(self, albedo=(0.5, 0.5, 0.5), texture=-1):
        return self._mat(capi.MAT_LAMBERTIAN, albedo=albedo, texture=texture)

    def metal(self, albedo, fuzziness, texture=-1):
        f = F(fuzziness)
        f = (f if f >= 0 else F(0)) if f < 1 else F(1)          # Materials.cpp:96
        return self._mat(capi.MAT_METAL, albedo=albedo, fuzz=float(f), texture=texture)

    def dielectric(self, ior):
        return self._mat(capi.MAT_DIELECTRIC, ior=ior)

    def diffuse_light(self, intensity):
        return self._mat(capi.MAT_DIFFUSE_LIGHT, emit=intensity)

    # ---- objects (SimpleObject.cpp) --------------------------------------------------------
    def _prim(self, kind, off, geom, axes, rot, mat, chain=None, center=(0.0, 0.0, 0.0)):
        self.kind.append(kind); self.off.append(off); self.geom.append(geom); self.axes.append(axes)
        self.rot.append(rot); self.pmat.append(mat)
        self.chain.append(chain); self.center.append(tuple(float(x) for x in center))
        self._arrays = None

    # ---- general hitables (Hitables.h:20-106 compose freely) -------------------------------
    def hitable(self, tree, mat, force_chain=False):
        """One Object whose m_hitable is `tree`, a nesting of
            ("sphere", center, radius)                                  SphereHitable              Hitables.cpp:6-11
            ("rect", aAxis, bAxis, a0, a1, b0, b1, c, reverse)          AxisAlignedRectHitable     :80-91
            ("translate", child, offset)                                TranslatedInstance         :163-173
            ("rotate", child, angle_radians, axis)                      RotatedInstance            :182-264
            ("combo", [children])                                       HitableCombo               :130-142
        A combo distributes over its children (each leaf keeps the instances above it, outermost first), so the object becomes a
        run of primitives in the combo's own order.  Leaves whose chain is one of the factories' shapes -- Translate(Sphere at 0),
        Translate(RotZ(RotY(RotX(Rect)))) -- are stored in the factory form unless force_chain; every other leaf in chain form."""
        self.obj_first.append(len(self.kind)); n0 = len(self.kind)

        def walk(node, ops, force):
            tag = node[0]
            if tag == "translate":
                walk(node[1], ops + [(capi.OP_TRANSLATE, tuple(float(F(x)) for x in node[2]))], force)
            elif tag == "rotate":
                a = F(node[2])
                walk(node[1], ops + [(capi.OP_ROTATE_X + int(node[3]), (float(F(math.sin(float(a)))), float(F(math.cos(float(a)))), 0.0))], force)
            elif tag == "combo":
                # a combo of rects alone is a cube's (its faces share the chain and may keep the factory form); any other combo
                # hands its leaves over in chain form
                force = force or any(ch[0] != "rect" for ch in node[1])
                for ch in node[1]:
                    walk(ch, ops, force)
            elif tag == "sphere":
                c = tuple(float(F(x)) for x in node[1]); r = float(F(node[2]))
                if not force and len(ops) == 1 and ops[0][0] == capi.OP_TRANSLATE and c == (0.0, 0.0, 0.0):
                    self._prim(capi.PRIM_SPHERE, ops[0][1], (r, 0, 0, 0, 0), 0, (0, 1, 0, 1, 0, 1), mat)
                else:
                    self._leaf_chain(capi.PRIM_SPHERE, (r, 0, 0, 0, 0), 0, ops, mat, c)
            elif tag == "rect":
                _, a, b, a0, a1, b0, b1, cc, rev = node
                a0, a1, b0, b1, cc = (float(F(x)) for x in (a0, a1, b0, b1, cc))
                if a0 > a1: a0, a1 = a1, a0                       # Hitables.cpp:86-90
                if b0 > b1: b0, b1 = b1, b0
                kinds = [k for k, _ in ops]
                if not force and kinds == [capi.OP_TRANSLATE, capi.OP_ROTATE_Z, capi.OP_ROTATE_Y, capi.OP_ROTATE_X]:
                    rot = (ops[3][1][0], ops[3][1][1], ops[2][1][0], ops[2][1][1], ops[1][1][0], ops[1][1][1])
                    self._prim(capi.PRIM_RECT, ops[0][1], (a0, a1, b0, b1, cc), capi.axes_pack(a, b, 3 - a - b, rev, 1), rot, mat)
                elif not force and kinds in ([], [capi.OP_TRANSLATE]):           # a bare or merely translated rect
                    self._prim(capi.PRIM_RECT, ops[0][1] if ops else (0.0, 0.0, 0.0), (a0, a1, b0, b1, cc), capi.axes_pack(a, b, 3 - a - b, rev, 0), (0, 1, 0, 1, 0, 1), mat)
                else:
                    self._leaf_chain(capi.PRIM_RECT, (a0, a1, b0, b1, cc), capi.axes_pack(a, b, 3 - a - b, rev, 0), ops, mat, (0.0, 0.0, 0.0))
            else:
                raise ValueError(f"unknown hitable {tag!r}")

        walk(tree, [], force_chain)
        if len(self.kind) == n0:
            raise ValueError("hitable without a leaf")
        self.obj_count.append(len(self.kind) - n0)

    def _leaf_chain(self, kind, geom, axes, ops, mat, center):
        if len(ops) > capi.MAX_CHAIN_OPS:
            raise ValueError(f"an instance chain of {len(ops)} ops (the device takes {capi.MAX_CHAIN_OPS})")
        self._prim(kind, (0.0, 0.0, 0.0), geom, axes, (0, 1, 0, 1, 0, 1), mat, chain=list(ops), center=center)

    def sphere(self, center, radius, mat):                       # SimpleObject.cpp:129-139
        self.obj_first.append(len(self.kind)); self.obj_count.append(1)
        self._prim(capi.PRIM_SPHERE, center, (radius, 0, 0, 0, 0), 0, (0, 1, 0, 1, 0, 1), mat)

    @staticmethod
    def _rot6(rotation):
        r = [F(x) for x in rotation]
        return tuple(float(f(x)) for x in r for f in (lambda a: F(math.sin(float(a))), lambda a: F(math.cos(float(a)))))

    def rect(self, axes, center, rotation, width, height, reverse, mat):   # SimpleObject.cpp:141-184
        a, b = {XY_RECT: (0, 1), XZ_RECT: (0, 2), ZY_RECT: (2, 1)}[axes]
        c = 3 - a - b                                            # Hitables.cpp:84
        ha, hb = F(width) / F(2.0), F(height) / F(2.0)
        self.obj_first.append(len(self.kind)); self.obj_count.append(1)
        self._prim(capi.PRIM_RECT, center, (-ha, ha, -hb, hb, 0.0), capi.axes_pack(a, b, c, reverse, 1), self._rot6(rotation), mat)

    def cube(self, center, rotation, size, mat):                 # SimpleObject.cpp:249-279
        s = np.asarray(size, F)
        mn = (F(0) - s) * F(0.5); mx = s * F(0.5)
        faces = [(0, 2, mn[0], mx[0], mn[2], mx[2], mx[1], 0), (0, 2, mn[0], mx[0], mn[2], mx[2], mn[1], 1),
                 (0, 1, mn[0], mx[0], mn[1], mx[1], mx[2], 0), (0, 1, mn[0], mx[0], mn[1], mx[1], mn[2], 1),
                 (2, 1, mn[2], mx[2], mn[1], mx[1], mx[0], 0), (2, 1, mn[2], mx[2], mn[1], mx[1], mn[0], 1)]
        self.obj_first.append(len(self.kind)); self.obj_count.append(6)
        rot = self._rot6(rotation)
        for (a, b, a0, a1, b0, b1, cc, rev) in faces:
            if a0 > a1: a0, a1 = a1, a0                           # Hitables.cpp:86-90
            if b0 > b1: b0, b1 = b1, b0
            self._prim(capi.PRIM_RECT, center, (a0, a1, b0, b1, cc), capi.axes_pack(a, b, 3 - a - b, rev, 1), rot, mat)

    def set_bvh(self, node_min, node_max, left, right, root):
        self.nodes = (np.ascontiguousarray(node_min, F).reshape(-1, 3), np.ascontiguousarray(node_max, F).reshape(-1, 3),
                      np.ascontiguousarray(left, np.int32), np.ascontiguousarray(right, np.int32), int(root))
        self._arrays = None

    # ---- flatten ---------------------------------------------------------------------------
    def arrays(self):
        if self._arrays is None:
            u32 = lambda x: np.ascontiguousarray(x, np.uint32)
            f32 = lambda x, k: np.ascontiguousarray(x, F).reshape(-1, k) if k > 1 else np.ascontiguousarray(x, F)
            self._arrays = dict(
                prim_kind=u32(self.kind), prim_off=f32(self.off, 3), prim_geom=f32(self.geom, 5), prim_axes=u32(self.axes),
                prim_rot=f32(self.rot, 6), prim_mat=u32(self.pmat), obj_first=u32(self.obj_first), obj_count=u32(self.obj_count),
                mat_kind=u32(self.mkind), mat_albedo=f32(self.albedo, 3), mat_fuzz=f32(self.fuzz, 1), mat_ior=f32(self.ior, 1),
                mat_emit=f32(self.emit, 3))
            if any(c is not None for c in self.chain):
                first, nops, kinds, args = [], [], [], []
                for c in self.chain:
                    if c is None:
                        first.append(0); nops.append(0)
                    else:
                        first.append(len(kinds) + 1); nops.append(len(c))
                        kinds += [k for k, _ in c]; args += [a for _, a in c]
                self._arrays.update(prim_chain=u32(first), prim_nops=u32(nops), prim_center=f32(self.center, 3),
                                    op_kind=u32(kinds if kinds else [0]), op_arg=f32(args if args else [(0.0, 0.0, 0.0)], 3))
                self._n_ops = len(kinds)
            if self.textures:
                sizes = np.array([t.shape[0] * t.shape[1] for t in self.textures], np.uint64)
                self._arrays.update(tex_width=u32([t.shape[1] for t in self.textures]), tex_height=u32([t.shape[0] for t in self.textures]),
                                    tex_offset=u32(np.concatenate([[0], np.cumsum(sizes)[:-1]])),
                                    tex_rgba8=np.concatenate([t.reshape(-1) for t in self.textures]),
                                    mat_texture=np.ascontiguousarray(self.mtex + [-1] * (len(self.mkind) - len(self.mtex)), np.int32))
        return self._arrays

    @property
    def n_prims(self):
        return len(self.kind)

    @property
    def n_objects(self):
        return len(self.obj_first)

    def as_struct(self):
        """A mirt_scene / orc_scene over this scene's arrays (which stay owned by `self`)."""
        import ctypes as C
        a = self.arrays()
        s = capi.SceneStruct()
        s.n_prims = len(self.kind); s.n_objects = len(self.obj_first); s.n_mats = len(self.mkind)
        for k, v in a.items():
            ct = {np.dtype(np.uint32): capi.u32p, np.dtype(np.float32): capi.f32p, np.dtype(np.uint8): capi.u8p, np.dtype(np.int32): capi.i32p}[v.dtype]
            setattr(s, k, v.ctypes.data_as(ct))
        if self.nodes is not None:
            mn, mx, l, r, root = self.nodes
            s.n_nodes = l.shape[0]; s.node_min = mn.ctypes.data_as(capi.f32p); s.node_max = mx.ctypes.data_as(capi.f32p)
            s.node_left = l.ctypes.data_as(capi.i32p); s.node_right = r.ctypes.data_as(capi.i32p); s.root = root
        else:
            s.n_nodes = 0; s.root = capi.CHILD_NONE
        s.ambient = (C.c_float * 3)(*self.ambient)
        s.n_textures = len(self.textures)
        s.n_ops = getattr(self, "_n_ops", 0) if "prim_chain" in a else 0
        return s


# ---- camera (SimpleCamera.cpp:32-53,139-191) ------------------------------------------------
def _dot(a, b):
    return F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2])


def _normalize(a):
    l = np.sqrt(_dot(a, a))
    return np.array([a[0] / l, a[1] / l, a[2] / l], F) if l != 0 else np.zeros(3, F)


def _cross(a, b):
    return np.array([F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0])], F)


def make_camera(look_from, look_at, fov_deg, aperture, aspect):
    origin, focus = np.asarray(look_from, F), np.asarray(look_at, F)
    vup = np.array([0, 1, 0], F)
    d = origin - focus
    focus_dist = np.sqrt(_dot(d, d))
    w = _normalize(d)
    u = _normalize(_cross(vup, w))
    v = _cross(w, u)
    theta = F(F(fov_deg) * F(math.pi)) / F(180.0)
    half_h = F(math.tan(float(theta / F(2.0)))) * focus_dist
    half_w = half_h * F(aspect)
    tl = ((origin - half_w * u) + half_h * v) - focus_dist * w
    cam = capi.CameraStruct()
    for name, val in (("origin", origin), ("top_left", tl), ("horizontal", (F(2.0) * half_w) * u),
                      ("vertical", (F(-2.0) * half_h) * v), ("u", u), ("v", v)):
        setattr(cam, name, (type(cam.origin))(*[float(x) for x in val]))
    cam.lens_radius = float(F(aperture) / F(2.0))
    return cam


# ---- worlds (World.cpp:25-101, Resouces.cpp:126-208) ----------------------------------------
def cornell_box(aspect=1.0):
    """WORLD_ID_CORNELL_BOX, World.cpp:69-101 (the three-ball variant that ships enabled)."""
    s = FlatScene(ambient=(0.0, 0.0, 0.0))
    bright = s.diffuse_light((15.0, 15.0, 15.0))          # MATERIAL_ID_LIGHTSOURCE_BRIGHT  Resouces.cpp:183
    lam4 = s.lambertian((0.12, 0.45, 0.15))               # MATERIAL_ID_LAMBERTIAN4         :151
    lam2 = s.lambertian((0.65, 0.05, 0.05))               # MATERIAL_ID_LAMBERTIAN2         :139
    lam3 = s.lambertian((0.73, 0.73, 0.73))               # MATERIAL_ID_LAMBERTIAN3         :145
    ball1 = s.lambertian((0.19, 0.2, 0.14))               # MATERIAL_ID_CORNELL_BALL1       :199
    glass = s.dielectric(1.5)                             # MATERIAL_ID_DIELECTRIC          :163
    ball2 = s.metal((0.26, 0.06, 0.06), 0.20)             # MATERIAL_ID_CORNELL_BALL2       :205-207
    z = (0.0, 0.0, 0.0)
    s.rect(XZ_RECT, (0.0, 0.99, 0.0), z, 0.65, 0.65, True, bright)     # World.cpp:72
    s.rect(ZY_RECT, (1.0, 0.0, 0.0), z, 2.0, 2.0, True, lam4)          # :75
    s.rect(ZY_RECT, (-1.0, 0.0, 0.0), z, 2.0, 2.0, False, lam2)        # :77
    s.rect(XY_RECT, (0.0, 0.0, -1.0), z, 2.0, 2.0, False, lam3)        # :79
    s.rect(XZ_RECT, (0.0, 1.0, 0.0), z, 2.0, 2.0, True, lam3)          # :80
    s.rect(XZ_RECT, (0.0, -1.0, 0.0), z, 2.0, 2.0, False, lam3)        # :81
    r = F(0.32)
    y = float(F(-1.0) + r)
    s.sphere((-0.38, y, -0.35), float(r), ball1)                       # :92
    s.sphere((0.0, y, 0.50), float(r), glass)                          # :93
    s.sphere((0.38, y, -0.35), float(r), ball2)                        # :94
    cam = make_camera((0.0, 0.64, 4.63), (0.0, -0.26, -1.0), 30.0, 0.0, aspect)   # :96
    return s, cam


def cornell_two_boxes(aspect=1.0):
    """The `#if 0` variant of World.cpp:83-89: two rotated SimpleObjectCubes instead of the balls."""
    s, _ = cornell_box(aspect)
    del_from = 6
    for lst in (s.kind, s.off, s.geom, s.axes, s.rot, s.pmat):
        del lst[del_from:]
    del s.obj_first[del_from:]; del s.obj_count[del_from:]
    s._arrays = None
    lam3 = 3
    h = F(1.2)
    s.cube((-0.3, float(F(-1.0) + h * F(0.5)), -0.4), (0.0, 0.3, 0.0), (0.6, float(h), 0.6), lam3)
    h = F(0.6)
    s.cube((0.32, float(F(-1.0) + h * F(0.5)), 0.4), (0.0, -0.3, 0.0), (float(h),) * 3, lam3)
    cam = make_camera((0.0, 0.0, 4.0), (0.0, 0.0, 0.0), 40.0, 0.0, aspect)
    return s, cam


def chains_world(aspect=1.0):
    """WORLD_ID_CHAINS of the C++ layer (host.cpp World::ConstructWorld): the hitable vocabulary beyond the factories' shapes."""
    s = FlatScene(ambient=(0.6, 0.7, 0.9))
    grey = s.lambertian((0.5, 0.5, 0.5)); white = s.diffuse_light((4.0, 4.0, 4.0)); metal = s.metal((0.7, 0.6, 0.5), 0.0)
    brown = s.lambertian((0.4, 0.2, 0.1)); red = s.lambertian((0.65, 0.05, 0.05)); glass = s.dielectric(1.5)
    s.rect(XZ_RECT, (0.0, -2.0, 0.0), (0.0, 0.0, 0.0), 12.0, 12.0, False, grey)
    s.rect(XZ_RECT, (0.0, 5.0, 0.0), (0.0, 0.2, 0.0), 3.0, 3.0, True, white)
    s.hitable(("translate", ("rotate", ("sphere", (1.0, 2.0, 3.0), 0.8), 30.0, 1), (0.25, -0.5, 0.125)), metal)
    s.hitable(("combo", [("sphere", (-1.5, 0.2, 0.3), 0.6), ("sphere", (-0.9, 0.5, 0.1), 0.5)]), brown)
    s.hitable(("rotate", ("translate", ("rotate", ("rect", 0, 2, -0.8, 0.8, -0.6, 0.6, 0.1, False), 0.7, 2), (0.4, -1.0, 0.9)), -0.5, 0), red)
    pair = ("combo", [("sphere", (0.5, 0.0, 0.0), 0.25), ("translate", ("sphere", (0.0, 0.0, 0.0), 0.2), (-0.5, 0.1, 0.0))])
    every = ("combo", [("sphere", (0.0, 0.0, 0.0), 0.4), ("translate", ("rect", 0, 1, -0.3, 0.3, -0.3, 0.3, 0.0, False), (0.0, 0.6, 0.0)), pair])
    s.hitable(("translate", ("rotate", every, 1.1, 1), (2.2, -1.0, 0.5)), glass)
    cam = make_camera((0.5, 1.0, 9.0), (0.0, 0.0, 0.0), 45.0, 0.05, aspect)
    return s, cam


def textured_world(aspect, soil_rgba, checker_rgba):
    """WORLD_ID_TEXTURED of the host layer: the two image-textured materials of Resouces.cpp:166-176 on a floor,
    a back wall and a ball.  soil_rgba / checker_rgba: decoded texels, (h, w, 4) uint8, row 0 = top."""
    s = FlatScene(ambient=(0.5, 0.6, 0.8))
    soil = s.lambertian(texture=s.texture(soil_rgba))                   # Lambertian(texture)
    checker = s.metal((0.5, 0.5, 0.5), 0.7, texture=s.texture(checker_rgba))   # Metal(texture, 0.7f)
    white = s.diffuse_light((4.0, 4.0, 4.0))                            # MATERIAL_ID_LIGHTSOURCE_WHITE
    s.rect(XZ_RECT, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 4.0, 4.0, False, soil)
    s.rect(XY_RECT, (0.0, 1.5, -2.0), (0.0, 0.0, 0.0), 4.0, 3.0, False, soil)
    s.sphere((0.0, 0.7, 0.0), 0.7, checker)
    s.rect(XZ_RECT, (0.0, 3.0, 0.0), (0.0, 0.0, 0.0), 1.5, 1.5, True, white)
    cam = make_camera((0.0, 1.5, 5.0), (0.0, 0.7, 0.0), 35.0, 0.0, aspect)
    return s, cam


class _HostRng:
    """Deterministic host draws for scene construction: Philox4x32-10 keyed by `seed`, counter =
    draw index; mapped like Randomizer::RandomUNorm (Randomizer.h:9-17,29-32)."""

    def __init__(self, seed):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.i = 0
        self.buf = []

    @staticmethod
    def philox(ctr, key):
        c = list(ctr); k = list(key)
        for _ in range(10):
            p0 = 0xD2511F53 * c[0]; p1 = 0xCD9E8D57 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
            k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
        return c

    def unorm(self):
        if not self.buf:
            self.buf = self.philox((self.i, 0, 0, 0x5CE9E), self.key)
            self.i += 1
        return F(self.buf.pop(0) / 4294967296.0)


def _philox_np(counters, key):
    """Philox4x32-10 over an (n,4) uint32 counter array (vectorised twin of _HostRng.philox)."""
    c = [counters[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]; p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32; k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=1).astype(np.uint32)


def _host_draws(seed, first, count):
    """Raw draws [first, first+count) of the _HostRng(seed) stream, vectorised."""
    b0, b1 = first // 4, (first + count + 3) // 4
    ctr = np.zeros((b1 - b0, 4), np.uint32)
    ctr[:, 0] = np.arange(b0, b1, dtype=np.uint64).astype(np.uint32); ctr[:, 3] = 0x5CE9E
    raw = _philox_np(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)
    return raw[first - 4 * b0: first - 4 * b0 + count]


def spheres_world(aspect=1.5, small_spheres=True, seed=1, grid=11, look_from=(11.0, 2.0, 3.0)):
    """WORLD_ID_RANDOM_SPHERES, World.cpp:25-67.  small_spheres=False is the `#if 1` block at :36
    disabled (BASELINE config 1, the 4-sphere scene); grid=11 is the reference's a,b in [-11,11),
    larger grids extend the same generator (BASELINE config 4: grid=500 -> ~1M spheres, look_from raised).
    The reference draws positions and the 50+50 random materials from std::random_device; here they come
    from Philox(seed) in the reference's draw order (Resouces.cpp:110-124 first, then World.cpp:41-54:
    chooseMat, x jitter, z jitter, and a material pick only for kept non-glass spheres)."""
    g = _HostRng(seed)
    s = FlatScene(ambient=(0.85, 0.9, 1.0))                            # World.cpp:63
    rl = [s.lambertian((float(g.unorm() * g.unorm()), float(g.unorm() * g.unorm()), float(g.unorm() * g.unorm()))) for _ in range(50)]
    rm = []
    for _ in range(50):
        alb = tuple(float(F(0.5) * (F(1) + g.unorm())) for _ in range(3))
        rm.append(s.metal(alb, float(F(0.5) * g.unorm())))
    lam0 = s.lambertian((0.5, 0.5, 0.5)); lam1 = s.lambertian((0.4, 0.2, 0.1))
    metal = s.metal((0.7, 0.6, 0.5), 0.0); glass = s.dielectric(1.5)
    assert (rl[0], rm[0], lam0, glass) == (0, 50, 100, 103)
    s.sphere((0.0, -1000.0, -0.0), 1000.0, lam0)                       # :28
    s.sphere((-4.0, 1.0, 0.0), 1.0, lam1)                              # :31
    s.sphere((0.0, 1.0, 0.0), 1.0, glass)                              # :32
    s.sphere((4.0, 1.0, 0.0), 1.0, metal)                              # :33
    if small_spheres:
        # The material pick consumes a draw only for kept non-glass spheres, so a cell's draws depend on all
        # earlier cells.  Cells are processed in blocks: draw a generous window, then resolve offsets by a scan.
        ncell = (2 * grid) ** 2
        raw = _host_draws(seed, 500, 4 * ncell)                         # 500 = the palette draws above
        un = (raw.astype(np.float64) / 4294967296.0).astype(F)
        aa = np.repeat(np.arange(-grid, grid), 2 * grid).astype(F); bb = np.tile(np.arange(-grid, grid), 2 * grid).astype(F)
        cen = np.zeros((ncell, 3), F); mats = np.zeros(ncell, np.int64); keep = np.zeros(ncell, bool)
        pos = 0
        for i in range(ncell):                                          # scalar scan over cells, vector draws
            choose, ux, uz = un[pos], un[pos + 1], un[pos + 2]; pos += 3
            cx = aa[i] + F(0.9) * ux; cz = bb[i] + F(0.9) * uz
            dx = cx - F(4.0); dz = cz - F(0.0)
            if np.sqrt(F(F(dx * dx) + F(0.0)) + F(dz * dz)) > F(0.9):   # :43
                m = glass
                if choose < F(0.8):
                    m = int(F(0.0) + un[pos] * F(50)); pos += 1        # :49 (START + u*COUNT, truncated)
                elif choose < F(0.95):
                    m = int(F(50.0) + un[pos] * F(50)); pos += 1       # :54
                keep[i] = True; mats[i] = m; cen[i] = (cx, F(0.2), cz)
        k = int(keep.sum())
        base = len(s.kind)
        s.kind += [capi.PRIM_SPHERE] * k
        s.off += [tuple(float(v) for v in c) for c in cen[keep]]
        s.geom += [(0.2, 0, 0, 0, 0)] * k
        s.axes += [0] * k; s.rot += [(0, 1, 0, 1, 0, 1)] * k; s.pmat += [int(m) for m in mats[keep]]
        s.obj_first += list(range(base, base + k)); s.obj_count += [1] * k
        s._arrays = None
    cam = make_camera(look_from, (0.0, 0.6, 0.0), 20.0, 0.1, aspect)   # :64
    return s, cam
