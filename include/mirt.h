/*
 * mirt.h -- C ABI of the MI355X (gfx950) path tracer that replaces the render loop of
 * ByeDream/RayTracingInAWeekend's HomemadeRayTracer.
 *
 * The reference has no FFI: its seam is the C++ class HomemadeRayTracer
 * (RayTracer/RayTracer/HomemadeRayTracer.h:11-33).  This header is what a replacement shared
 * library exports and what the drop-in C++ host layer (raytracinginaweekend_amd/host/) calls.
 * Plain pointers and sizes only; no exceptions cross the boundary; every entry point returns
 * MIRT_OK (0) or a negative status and records a message retrievable with mirt_last_error().
 * File:line citations below are relative to /root/reference/RayTracer/RayTracer/.
 *
 * A context is bound to one GPU and is single-owner (not thread-safe), like the reference's
 * tracer that runs synchronously on the UI thread (RayTracer.cpp:251-254).
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_OK             0
#define MIRT_ERR_ARG       -1   /* bad argument / inconsistent scene                    */
#define MIRT_ERR_HIP       -2   /* a HIP runtime call failed (message has the detail)   */
#define MIRT_ERR_STATE     -3   /* call order violated (e.g. render before upload)      */
#define MIRT_ERR_NOMEM     -4

enum { MIRT_PRIM_SPHERE = 0, MIRT_PRIM_RECT = 1 };
/* Materials.h:7-15 (MaterialID) */
enum { MIRT_MAT_DIFFUSE_LIGHT = 0, MIRT_MAT_LAMBERTIAN = 1, MIRT_MAT_METAL = 2, MIRT_MAT_DIELECTRIC = 3 };
#define MIRT_CHILD_NONE ((int32_t)0x80000000)
/* prim_axes (rect): aAxis | bAxis<<2 | cAxis<<4 | reverseFace<<6 | hasRotationChain<<7 */
#define MIRT_AXES_PACK(a, b, c, rev, rot) ((uint32_t)((a) | ((b) << 2) | ((c) << 4) | ((rev) << 6) | ((rot) << 7)))

/*
 * Flattened scene: what World::ConstructWorld (World.cpp:15-110) leaves behind, as SoA arrays.
 *  - a primitive is one leaf hitable together with its instance chain:
 *      sphere = TranslatedInstance(SphereHitable(0,r))                        SimpleObject.cpp:136
 *      rect   = TranslatedInstance(RotZ(RotY(RotX(AxisAlignedRectHitable))))  SimpleObject.cpp:173-181
 *  - an object is one `Object` (SimpleObject.h:27-49): a run of primitives; SimpleObjectCube's
 *    HitableCombo (SimpleObject.cpp:261-276) is an object of six rect primitives
 *  - the BVH is the SimpleObjectBVHNode tree (SimpleObject.cpp:293-346), children >= 0 are nodes,
 *    children < 0 are objects (~child), MIRT_CHILD_NONE is a null child (SimpleObject.cpp:311)
 * All host pointers are borrowed for the duration of the call.
 */
typedef struct mirt_scene {
    uint32_t        struct_size; /* sizeof(mirt_scene) as the caller compiled it (see "ABI guard" below) */
    uint32_t        n_prims;
    const uint32_t *prim_kind;   /* MIRT_PRIM_*                                            */
    const float    *prim_off;    /* [3n] TranslatedInstance::m_offset, Hitables.h:87       */
    const float    *prim_geom;   /* [5n] sphere: radius ; rect: a0,a1,b0,b1,c Hitables.h:50 */
    const uint32_t *prim_axes;   /* rect: MIRT_AXES_PACK                                   */
    const float    *prim_rot;    /* [6n] sin,cos of the x,y,z RotatedInstance, Hitables.h:97-98 */
    const uint32_t *prim_mat;    /* material index                                         */
    uint32_t        n_objects;
    const uint32_t *obj_first;
    const uint32_t *obj_count;
    uint32_t        n_mats;
    const uint32_t *mat_kind;    /* MIRT_MAT_*                                             */
    const float    *mat_albedo;  /* [3n] ITexture2D::Sample of a single-colour texture     */
    const float    *mat_fuzz;    /* [n]  Metal::m_data.fuzziness.x   Materials.cpp:96      */
    const float    *mat_ior;     /* [n]  Dielectric refractiveIndex  Materials.cpp:169     */
    const float    *mat_emit;    /* [3n] DiffuseLight intensity      Materials.cpp:257     */
    uint32_t        n_nodes;     /* may be 0: the library then builds its own BVH          */
    const float    *node_min;    /* [3n] SimpleObjectBVHNode::m_bindingBox                 */
    const float    *node_max;    /* [3n]                                                   */
    const int32_t  *node_left;
    const int32_t  *node_right;
    int32_t         root;
    float           ambient[3];  /* LightSources::GetAmbientLight, LightSources.h:21       */
    /* image textures (SimpleTexture2D_TGAImage, SimpleTexture2D.cpp:109-147); n_textures may be 0 and
     * mat_texture NULL: every material then samples its single colour mat_albedo */
    uint32_t        n_textures;
    const uint32_t *tex_width;   /* [n]  m_width                                           */
    const uint32_t *tex_height;  /* [n]  m_height                                          */
    const uint32_t *tex_offset;  /* [n]  index of the texture's first texel in tex_rgba8/4 */
    const uint8_t  *tex_rgba8;   /* m_pixelData of all textures back to back: R,G,B,A per texel, row 0 = top
                                    (what tgaRead(.., TGA_READER_ABGR) returns, tga_reader.cpp) */
    const int32_t  *mat_texture; /* [n_mats] texture sampled for the albedo (Lambertian, Metal), -1 = mat_albedo */
    /* General instance chains.  The reference's hitables compose freely (Hitables.h:20-106): SphereHitable has its own
     * m_center (Hitables.cpp:20,37,65), TranslatedInstance / RotatedInstance nest in any order (Hitables.cpp:163-173,241-264)
     * and HitableCombo holds any IHitable (Hitables.cpp:130-142).  A primitive in CHAIN FORM is one leaf with the list of the
     * instances around it, outermost first; a combo distributes over its children, so an object is still a run of primitives,
     * tested in order with t_max shrinking.  prim_chain may be NULL (n_ops 0): every primitive is then in the factory form above. */
    const uint32_t *prim_chain;  /* [n_prims] 0 = factory form (prim_off, prim_rot, sphere centre 0); k > 0 = chain form, whose ops
                                    are op_kind/op_arg[k-1 .. k-1+prim_nops[p]) and which ignores prim_off, prim_rot and the rotation
                                    bit of prim_axes                                          */
    const uint32_t *prim_nops;   /* [n_prims] instances around the leaf, 0..MIRT_MAX_CHAIN_OPS */
    const float    *prim_center; /* [3 n_prims] SphereHitable::m_center of chain-form spheres  */
    uint32_t        n_ops;
    const uint32_t *op_kind;     /* [n_ops] MIRT_OP_*                                          */
    const float    *op_arg;      /* [3 n_ops] translate: m_offset (Hitables.h:87); rotate: m_sinTheta, m_cosTheta (Hitables.h:97-98), 0 */
} mirt_scene;
#define MIRT_MAX_CHAIN_OPS 8u
enum { MIRT_OP_TRANSLATE = 0, MIRT_OP_ROTATE_X = 1, MIRT_OP_ROTATE_Y = 2, MIRT_OP_ROTATE_Z = 3 };  /* 1 + RotatedInstance's rotateAxis */

/* The SimpleCamera members GetRay reads (SimpleCamera.h:35-56, SimpleCamera.cpp:55-60). */
typedef struct mirt_camera {
    float origin[3], top_left[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius;
} mirt_camera;

/* HitRecord, Hitables.h:9-17 (material pointer -> index; prim = winning primitive) */
typedef struct mirt_hit {
    float   t, pos[3], nrm[3], u, v;
    int32_t mat;
    int32_t prim;
} mirt_hit;

/* render flags */
#define MIRT_FLAG_ONE_SPP        1u  /* m_enable1SPP,          HomemadeRayTracer.cpp:114-121 */
#define MIRT_FLAG_NORMAL_DISPLAY 2u  /* m_enableNormalDisplay, HomemadeRayTracer.cpp:164-167 */
#define MIRT_FLAG_ACCEL_LIST     4u  /* closest hit by a linear pass over all objects instead
                                        of the BVH (what HitableCombo::Hit does, Hitables.cpp:130-142) */
#define MIRT_FLAG_ACCEL_AUTO     16u /* library picks LIST for scenes of up to 32 primitives, ORDERED otherwise */
#define MIRT_FLAG_ACCEL_ORDERED  32u /* walk the BVH nearer child first instead of the reference's
                                        left-then-right order (SimpleObject.cpp:394-402): same closest
                                        t, far fewer boxes; only the winner of an exact tie can differ.
                                        No accel flag = the reference's order on the tree as given. */

typedef struct mirt_render_params {
    uint32_t struct_size;        /* sizeof(mirt_render_params) as the caller compiled it   */
    uint32_t width, height;      /* image->m_width/m_height, HomemadeRayTracer.cpp:100-101 */
    uint32_t spp_begin;          /* first global sample index rendered by this call        */
    uint32_t spp_count;          /* SAMPLE_PER_PIXEL (:21) made a run-time range           */
    uint32_t max_depth;          /* MAX_SAMPLE_DEPTH (:22), reference value 50             */
    uint32_t flags;
    uint64_t seed;               /* Philox key; std::random_device has no seed (Randomizer.h:11) */
} mirt_render_params;

typedef struct mirt_stats {
    uint32_t struct_size;        /* sizeof(mirt_stats) as the caller compiled it: set it BEFORE the call (the
                                    library fills everything else and leaves this field alone) */
    uint64_t rays;               /* rays traced (exact: from the path-length histogram)    */
    uint64_t samples;            /* camera samples started                                 */
    uint64_t depth_hist[64];     /* rays per bounce index (0..63; deeper folded into 63)   */
    double   device_ms;          /* HIP-event time of the kernels of the call              */
    double   extend_ms;          /* with option time_launches=1: HIP-event time summed over the
                                    launches of the dominant kernel (engine 1: k_trace, one per
                                    batch; engine 0: the depth>=1 k_bounce launches)        */
    uint32_t bounce_launches;    /* number of tracing-kernel launches                      */
    uint32_t batches;
    double   first_ms;           /* engine 0 with time_launches=1: same for the depth-0 launches */
    uint32_t extend_launches;    /* launches timed into extend_ms                          */
    uint32_t first_launches;
    double   depth_ms[64];       /* engine 0 with time_launches=1: launch time per bounce index */
} mirt_stats;

typedef struct mirt_ctx mirt_ctx;

/* -- ABI guard ------------------------------------------------------------------------------ */
/* The structs above are filled on one side of the boundary and read on the other, so a host compiled against
 * another revision of this header must get an error code, not a stack overrun (no reference counterpart: the
 * reference is one executable).  Two layers:
 *  - mirt_scene, mirt_render_params and mirt_stats begin with `struct_size`; every entry point that takes one
 *    returns MIRT_ERR_ARG when it is not the size the library was built with (and writes nothing);
 *  - a host checks everything once at load: `if (MIRT_CHECK_ABI() != MIRT_OK) refuse to run`.
 * MIRT_ABI_VERSION is bumped whenever a struct, an enum value or a signature changes. */
#define MIRT_ABI_VERSION 3u
uint32_t mirt_abi_version(void);
int      mirt_check_abi(uint32_t version, size_t sizeof_scene, size_t sizeof_camera, size_t sizeof_hit,
                        size_t sizeof_render_params, size_t sizeof_stats);
#define MIRT_CHECK_ABI() mirt_check_abi(MIRT_ABI_VERSION, sizeof(mirt_scene), sizeof(mirt_camera), sizeof(mirt_hit), \
                                        sizeof(mirt_render_params), sizeof(mirt_stats))

/* -- lifecycle ------------------------------------------------------------------------------ */
/* Replaces `new HomemadeRayTracer(...)` + OnInit (RayTracer.cpp:129-130). */
int         mirt_create(int device_id, mirt_ctx **out);
void        mirt_destroy(mirt_ctx *ctx);                 /* HomemadeRayTracer::OnDestroy      */
const char *mirt_last_error(const mirt_ctx *ctx);        /* ctx may be NULL: last create error */
int         mirt_device_count(void);
/* Versions of the HIP runtime and driver this library is bound to in this process (hipRuntimeGetVersion /
 * hipDriverGetVersion; either pointer may be NULL).  No reference counterpart; hosts that hand the library device
 * pointers and streams made by another HIP user in the same process (bench.py: torch) record it beside the path of
 * the one libamdhip64 mapped. */
int         mirt_runtime_version(int *hip_runtime, int *hip_driver);
/* option names:
 *  "max_paths"      paths per batch (default 1 Gi; buffers are sized by the render at hand, up to this)
 *  "time_launches"  0/1: bracket every tracing-kernel launch with HIP events, sums reported in mirt_stats
 *  "engine"         1 (default): persistent kernel, one path per lane, no path queues;
 *                   0: wavefront formulation, one launch per bounce over compacted path queues.
 *                   Both produce the same bits.
 *                   (engine 0: single-colour scenes only; batches of at most 256 Mi paths)
 *  "blocks_per_cu"  workgroups per CU of the persistent grid (0, default: as many as fit for large launches, fewer for
 *                   launches of few paths per lane)
 *  "paths_per_lane" list walk of scenes without image textures: 1 (default) one path per lane; 2: two paths in every
 *                   lane, which share the passes that few lanes need (measured slower, kept under test).  Same image.
 *  "walk_yield"     near-first BVH walks in resumable slices: a wave leaves the walk loop once this many of its
 *                   lanes have finished, shades them and resumes the others (1..64); 0: every wave waits for its
 *                   longest walk; -1 (default): the library's choice, currently 52 (small trees) or 56 (trees whose top is read
 *                   from LDS, see lds_top) for batches of 2 Mi paths or more, else 0.  Same image.
 *  "lds_top"        sliced near-first walk: this many nodes of the top of the tree are read from LDS instead of memory
 *                   (0..256; -1, default: 32 for trees of 65536 nodes and more, else 0).  Takes effect at the next
 *                   mirt_upload_scene.  Same image.
 *  "leaf_wait"      near-first BVH walks: a lane that reaches an object waits until this many lanes of its wave wait
 *                   with it (or none has a node left) before the objects are tested, all in one pass (1..64; -1, default: 8 for
 *                   the sliced walk of a small tree, else 4).
 *                   Same image: a waiting lane does nothing in between.
 *  "tile_paths"     1 (default): with a BVH walk, the persistent engine hands out pixels in 8x8 tiles rather than
 *                   rows (compact patches of camera rays); 0: rows.  No effect on the image.
 *  "lds_budget", "stack_lds", "lds_mats", "lds_prims"  LDS plan of the next mirt_upload_scene: bytes per workgroup
 *                   up to which the top of the BVH is staged (default 4096), traversal-stack levels kept in LDS
 *                   (default 5, the rest spills to HBM), largest material table staged (default 1024 bytes),
 *                   largest primitive table staged (default 4096 bytes; such scenes are walked as a list under
 *                   MIRT_FLAG_ACCEL_AUTO when they have at most 32 primitives) */
int         mirt_set_option(mirt_ctx *ctx, const char *name, int64_t value);

/* -- scene ---------------------------------------------------------------------------------- */
/* Replaces the pointer graph TraceRay walks through World::GetObjectBVHTree()/GetLightSources()
 * (World.h:28,31).  Limits: a BVH of at most 60 levels, fewer than 2^25 nodes and fewer than 2^26 objects (the walks address
 * node and leaf records by signed 32-bit byte offsets); MIRT_ERR_ARG beyond. */
int mirt_upload_scene(mirt_ctx *ctx, const mirt_scene *scene);
/* Host-only, needs no GPU: the library's own BVH builder (binned SAH), the one mirt_upload_scene runs for a
 * scene without a tree.  The reference's counterpart is the SimpleObjectBVHNode constructor
 * (SimpleObject.cpp:293-346: random axis, median split, O(N log^2 N) vector copies).  Arrays must hold
 * capacity_nodes entries (2 * n_objects always suffices); any tree in `scene` is ignored. */
int mirt_build_bvh(const mirt_scene *scene, uint32_t capacity_nodes, float *node_min, float *node_max,
                   int32_t *node_left, int32_t *node_right, int32_t *root, uint32_t *n_nodes, uint32_t *depth);
/* Replaces the `const SimpleCamera*` argument of TraceRay (HomemadeRayTracer.h:22). */
int mirt_set_camera(mirt_ctx *ctx, const mirt_camera *cam);

/* -- render --------------------------------------------------------------------------------- */
/* HomemadeRayTracer::TraceRay (HomemadeRayTracer.cpp:90-154), device-resident form.
 * Adds samples [spp_begin, spp_begin+spp_count) of every pixel, in sample order, to d_sum_rgb
 * (DEVICE pointer, width*height*3 floats, row 0 = top, caller zero-initialises; with
 * MIRT_FLAG_ONE_SPP the buffer is overwritten).  `stream` is a hipStream_t (NULL = default
 * stream).  Synchronises the stream before returning when stats != NULL. */
int mirt_render_device(mirt_ctx *ctx, const mirt_render_params *p, float *d_sum_rgb,
                       void *stream, mirt_stats *stats);
/* mirt_render_device for callers that cannot allocate.  No counterpart in the reference (its TraceRay new[]s its
 * own buffer, HomemadeRayTracer.cpp:103): this entry point is for code under HIP-graph capture, tight serving loops
 * and harnesses that forbid hipMalloc.  Same kernels, same arguments (no statistics), same bits in d_sum_rgb; the
 * radiance slots, claim counters and the HBM part of the traversal stacks live in a fixed 56 MiB array in the
 * library's device image (one copy per GPU, held until the process ends) instead of hipMalloc'ed buffers, and the
 * call makes no hipMalloc / hipFree / allocating library call.  Limits (MIRT_ERR_ARG with a message beyond them):
 * batches of 2 Mi paths, at most 3971 batches per call (width*height*spp_count <= ~8.3e9), 31 MiB of stack spill
 * (always enough for a list walk; a BVH deeper than the LDS stack part needs (depth+1-stack_lds) * 4 bytes per
 * resident lane).  Persistent engine only.  Every call on a GPU shares that GPU's array: calls that overlap on one
 * GPU (two streams, two contexts, two threads) race -- serialise them. */
int mirt_render_device_noalloc(mirt_ctx *ctx, const mirt_render_params *p, float *d_sum_rgb, void *stream);
/* The tail of TraceRay + OutputImage::Render (HomemadeRayTracer.cpp:134-138, OutputImage.cpp:62-88):
 * mean over spp_total (skipped for ONE_SPP), per-channel sqrt, clamp, (UINT32)(c*255.0f)&0xFF, A=255.
 * d_pixels (device, w*h*3 post-gamma floats) and d_rgba8 (device, w*h*4) may each be NULL. */
int mirt_resolve_device(mirt_ctx *ctx, const float *d_sum_rgb, uint32_t width, uint32_t height,
                        uint32_t spp_total, uint32_t flags, float *d_pixels, uint8_t *d_rgba8,
                        void *stream);
/* Whole TraceRay with host outputs (each nullable): sum of radiance (w*h*3), post-gamma float
 * pixels (w*h*3, what TraceRay hands to OutputImage::Render) and RGBA8 (w*h*4, OutputImage::m_data). */
int mirt_render(mirt_ctx *ctx, const mirt_render_params *p, float *h_sum_rgb, float *h_pixels,
                uint8_t *h_rgba8, mirt_stats *stats);

/* The interactive caller one step above TraceRay (the viewer re-invokes TraceRay from its paint
 * loop, RayTracer.cpp:251-254): successive calls refine ONE image.  The radiance sum stays on the device between
 * calls; a call with spp_begin == 0 restarts it, any other call must continue where the previous one ended on the
 * same image size (else MIRT_ERR_STATE).  Outputs (each may be NULL) are the image of all samples so far; after
 * calls covering [0,a) and [a,b) they equal, bit for bit, those of one mirt_render of [0,b). */
int mirt_render_progressive(mirt_ctx *ctx, const mirt_render_params *p, float *h_sum_rgb, float *h_pixels_rgb,
                            uint8_t *h_rgba8, mirt_stats *stats);

/* -- multi-GPU: one render sharded over the GPUs of a node, inside one process ---------------- */
/* The reference parallelises TraceRay over image rows with OpenMP threads and no communication
 * (HomemadeRayTracer.cpp:107-110); it has no multi-device notion.  Here device g of G renders the global sample indices
 * [g*spp/G, (g+1)*spp/G) of every pixel -- the Philox counter holds the global index, so the sample set does not
 * depend on G -- with one host thread per device, the G radiance-sum buffers are summed onto device_ids[0] (the one
 * exchange of the path) and that device resolves.  The image equals the one-device image up to the re-association of
 * the per-pixel float sum.  (Processes that own one GPU each do the same with mirt_render_device + their own
 * collective: bench.py, one rank per GPU, RCCL through torch.distributed.) */
#define MIRT_MULTI_MAX_DEVICES 16
#define MIRT_REDUCE_AUTO 0   /* RCCL when the devices are distinct and librccl.so loads, else PEER                       */
#define MIRT_REDUCE_PEER 1   /* device 0 pulls every peer's buffer (hipMemcpyPeerAsync, one xGMI link each, side by side) and
                                adds them in rank order: a fixed association, bit-reproducible for a given G; a device may be
                                listed twice (contexts then share it) */
#define MIRT_REDUCE_RCCL 2   /* one ncclReduce(sum, float32, root 0) per device in a group; distinct devices only          */
typedef struct mirt_multi mirt_multi;
typedef struct mirt_multi_info {
    uint32_t struct_size;        /* sizeof(mirt_multi_info), set by the caller                                             */
    uint32_t n_devices;
    uint32_t reduce_kind;        /* MIRT_REDUCE_PEER / _RCCL as used; 0 with one device                                    */
    uint32_t pad;
    double   render_ms;          /* wall clock: all devices rendering (slowest one)                                        */
    double   reduce_ms;          /* wall clock: the sum onto device 0                                                      */
    double   resolve_ms;         /* wall clock: resolve + copies to the host outputs                                       */
    uint64_t reduce_bytes;       /* size of one radiance-sum buffer                                                        */
    int32_t  device[MIRT_MULTI_MAX_DEVICES];
    uint32_t spp_begin[MIRT_MULTI_MAX_DEVICES], spp_count[MIRT_MULTI_MAX_DEVICES];
    double   device_ms[MIRT_MULTI_MAX_DEVICES];   /* HIP-event time of each device's kernels                               */
} mirt_multi_info;
int         mirt_multi_create(const int *device_ids, int n, mirt_multi **out);
void        mirt_multi_destroy(mirt_multi *m);
const char *mirt_multi_last_error(const mirt_multi *m);               /* m may be NULL: last create error */
int         mirt_multi_device_count(const mirt_multi *m);
/* "reduce": MIRT_REDUCE_*; every other name goes to mirt_set_option of each device's context */
int         mirt_multi_set_option(mirt_multi *m, const char *name, int64_t value);
int         mirt_multi_upload_scene(mirt_multi *m, const mirt_scene *scene);     /* replicated on every device */
int         mirt_multi_set_camera(mirt_multi *m, const mirt_camera *cam);
/* rank's share of the samples [spp_begin, spp_begin+spp_count): the ranges of ranks 0..world-1 tile it in order */
void        mirt_multi_spp_range(uint32_t rank, uint32_t world, uint32_t spp_begin, uint32_t spp_count, uint32_t *begin, uint32_t *count);
/* Whole TraceRay over all devices; outputs as mirt_render (host pointers, each nullable).  stats: sums over the
 * devices (device_ms/extend_ms: the slowest).  info (nullable): who rendered what and what the exchange cost. */
int         mirt_multi_render(mirt_multi *m, const mirt_render_params *p, float *h_sum_rgb, float *h_pixels, uint8_t *h_rgba8,
                              mirt_stats *stats, mirt_multi_info *info);

/* -- function-level entry points (parity tests bind these; host pointers) -------------------- */
/* AABB::Hit, AABB.cpp:5-21: out[i*nbox+b] = ray i vs box b */
int mirt_aabb_hit_batch(mirt_ctx *ctx, const float *bmin, const float *bmax, size_t nbox,
                        const float *org, const float *dir, size_t nray,
                        float t_min, float t_max, uint8_t *out);
/* IHitable::Hit of one primitive's instance chain, Hitables.cpp:13-54,93-115,163-173,241-264 */
int mirt_prim_hit_batch(mirt_ctx *ctx, uint32_t prim, const float *org, const float *dir,
                        size_t nray, float t_min, float t_max, mirt_hit *out);
/* SimpleObjectBVHNode::Hit, SimpleObject.cpp:389-406 (flags: MIRT_FLAG_ACCEL_LIST for the list).  With an accel flag
 * (LIST / AUTO / ORDERED) t_min must be > 0, as on the render path (0.001): those walks cull on the sign of the
 * roots; MIRT_ERR_ARG otherwise.  Without one (the reference's order) any t_min is taken. */
int mirt_world_hit_batch(mirt_ctx *ctx, const float *org, const float *dir, size_t nray,
                         float t_min, float t_max, uint32_t flags, mirt_hit *out);
/* IMaterial::Scatter, Materials.cpp:19-33,99-106,172-217,260-264, with the raw 32-bit draws
 * (`rd()` of Randomizer.h:14,23) supplied: draws[i*draws_per_ray + k] */
int mirt_scatter_batch(mirt_ctx *ctx, const mirt_hit *hits, const float *dir_in, size_t n,
                       const uint32_t *draws, uint32_t draws_per_ray,
                       uint8_t *out_alive, float *out_atten, float *out_org, float *out_dir,
                       float *out_emit, uint32_t *consumed);
/* SimpleCamera::GetRay, SimpleCamera.cpp:55-60, with the raw draws of RandomInUnitDisk supplied */
int mirt_camera_rays(mirt_ctx *ctx, const float *uv, size_t n,
                     const uint32_t *draws, uint32_t draws_per_ray,
                     float *out_org, float *out_dir, uint32_t *consumed);
/* The RNG source: Philox4x32-10 blocks, out[4*i..] = philox(ctr[4*i..], key) (device-computed) */
int mirt_philox_batch(mirt_ctx *ctx, const uint32_t *ctr, size_t n, const uint32_t key[2], uint32_t *out);

/* Arithmetic self-test of the device's IEEE building blocks (no reference counterpart; parity
 * tests bind it): op 0 x/y, 1 sqrt(x), 2 x^5 as Schlick evaluates it (Optics.h:31), 3 the [0,1)
 * mapping and 4 the [-1,1] mapping of Randomizer.h:9-26 applied to the raw bits of x, 5 the raw shared-denominator
 * division sequence of the rect test (the hardware's f32 division with the denominator-only part hoisted; it equals
 * x/y bit for bit only inside its operand range and has NO fallback of its own -- the render kernel checks the range
 * per ray and takes plain divides outside it), 6 whether (x, y) lies inside that range (1.0) or not (0.0),
 * 7 the render kernels' square root (the lean correctly rounded form when a whole wave's operands lie in [2^-60, 2^60],
 * the general expansion otherwise), 8 the lean form alone (promised only inside that range), 9 the unsigned quotient
 * bits(x) / bits(y) through the host-made reciprocal the render kernels split path ids with (result as bits).
 * y is required for ops 0, 5, 6 and 9. */
int mirt_arith_batch(mirt_ctx *ctx, uint32_t op, const float *x, const float *y, size_t n, float *out);
/* SimpleTexture2D_TGAImage::Sample, SimpleTexture2D.cpp:130-147: uv = n (u,v) pairs -> n RGB triples of texture `tex` */
int mirt_tex_sample_batch(mirt_ctx *ctx, uint32_t tex, const float *uv, size_t n, float *rgb_out);

/* -- host utilities -------------------------------------------------------------------------- */
/* PPMImageMaker::OutputRGBA8ToFile, PPMImageMaker.cpp:9-48 (ASCII P3, alpha dropped) */
int mirt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgba8);
/* Formats either side of the reference's P3 (PPMImageMaker.cpp:31-43 writes ~12 bytes of
 * text per pixel, 200 MB at 4096^2).  P3/P6 take rgba8 (alpha dropped), PFM takes rgb_float: width*height*3
 * floats, row 0 = top as everywhere in this ABI (e.g. the gamma-corrected pixels of mirt_resolve_device or
 * the linear mean); the file is written bottom row first, little-endian ("PF", scale -1.0), as PFM defines. */
#define MIRT_IMG_P3  0
#define MIRT_IMG_P6  1
#define MIRT_IMG_PFM 2
int mirt_write_image(const char *path, int format, uint32_t width, uint32_t height, const uint8_t *rgba8, const float *rgb_float);

#ifdef __cplusplus
}
#endif
#endif
