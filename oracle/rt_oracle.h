/*
 * rt_oracle.h -- CPU restatement ("oracle") of the reference's path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call anything under oracle/.  The
 * product path (raytracinginaweekend_amd/, include/mirt.h) never links or calls it.
 *
 * PARITY UNPINNED: the reference (ByeDream/RayTracingInAWeekend) holds no tests, golden
 * vectors or fixtures for this path, and it cannot be built here:
 * every arithmetic operation forwards to DirectXMath (Windows SDK 10.0.16299, absent) and
 * the sources include <windows.h>/<d3d12.h>.  Writing stand-ins for those headers is not
 * allowed, so no oracle/_ref build exists.  This file restates the reference algorithm by
 * reading its sources; the DirectXMath SSE2 operation order is restated from its published
 * behaviour: XMVector3Dot = (x*x' + y*y') + z*z'; XMVectorScale = per-lane multiply;
 * XMVector3Normalize = v / sqrt(dot) with zero -> 0 and inf -> NaN; XMVectorNegate = 0 - v;
 * XMVectorClamp = min(max(v, lo), hi) with SSE maxps/minps NaN behaviour; no FMA contraction
 * (MSVC /fp:precise).  Build with -ffp-contract=off.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/RayTracer/RayTracer/).
 *
 * Data interchange: the flattened scene below is layout-identical to `mirt_scene` of
 * include/mirt.h so that a test hands the very same arrays to both sides.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flattened scene (same layout as mirt_scene) ------------------------------------ */

enum { ORC_PRIM_SPHERE = 0, ORC_PRIM_RECT = 1 };
enum { ORC_MAT_DIFFUSE_LIGHT = 0, ORC_MAT_LAMBERTIAN = 1, ORC_MAT_METAL = 2, ORC_MAT_DIELECTRIC = 3 }; /* Materials.h:7-15 */
#define ORC_CHILD_NONE ((int32_t)0x80000000)

/* prim_axes bit layout (rect only): aAxis | bAxis<<2 | cAxis<<4 | reverseFace<<6 | hasRotationChain<<7 */
#define ORC_AXES_PACK(a, b, c, rev, rot) ((uint32_t)((a) | ((b) << 2) | ((c) << 4) | ((rev) << 6) | ((rot) << 7)))

typedef struct orc_scene {
    uint32_t        struct_size; /* unused here: keeps the layout identical to mirt_scene (one Python mirror serves both) */
    /* primitives, SoA, n_prims entries */
    uint32_t        n_prims;
    const uint32_t *prim_kind;   /* ORC_PRIM_*                                              */
    const float    *prim_off;    /* [3n] TranslatedInstance::m_offset                       */
    const float    *prim_geom;   /* [5n] sphere: radius,-,-,-,- ; rect: a0,a1,b0,b1,c       */
    const uint32_t *prim_axes;   /* rect: ORC_AXES_PACK                                     */
    const float    *prim_rot;    /* [6n] sin,cos of the x, y, z RotatedInstance (if rot bit) */
    const uint32_t *prim_mat;    /* material index                                          */
    /* objects (BVH leaves; a cube is one object of 6 rect prims = HitableCombo) */
    uint32_t        n_objects;
    const uint32_t *obj_first;
    const uint32_t *obj_count;
    /* materials, SoA */
    uint32_t        n_mats;
    const uint32_t *mat_kind;    /* ORC_MAT_*          */
    const float    *mat_albedo;  /* [3n]               */
    const float    *mat_fuzz;    /* [n]  Metal         */
    const float    *mat_ior;     /* [n]  Dielectric    */
    const float    *mat_emit;    /* [3n] DiffuseLight  */
    /* BVH (SimpleObjectBVHNode tree flattened); child >= 0: node, child < 0: object ~child */
    uint32_t        n_nodes;
    const float    *node_min;    /* [3n] */
    const float    *node_max;    /* [3n] */
    const int32_t  *node_left;
    const int32_t  *node_right;
    int32_t         root;        /* node index, or ~object when the scene is one object     */
    float           ambient[3];  /* LightSources::GetAmbientLight                           */
    /* image textures (SimpleTexture2D_TGAImage); n_textures 0 / mat_texture NULL = single colours only */
    uint32_t        n_textures;
    const uint32_t *tex_width, *tex_height, *tex_offset;   /* [n] size and first texel index            */
    const uint8_t  *tex_rgba8;   /* R,G,B,A per texel, row 0 = top                                     */
    const int32_t  *mat_texture; /* [n_mats] texture index or -1                                       */
    /* general instance chains (include/mirt.h): chain-form primitives = a leaf + its instances, outermost first */
    const uint32_t *prim_chain;  /* [n_prims] or NULL; 0 = factory form, k > 0 = ops [k-1, k-1+prim_nops[p])          */
    const uint32_t *prim_nops;
    const float    *prim_center; /* [3n] SphereHitable::m_center (chain form)                                          */
    uint32_t        n_ops;
    const uint32_t *op_kind;     /* 0 translate, 1..3 rotate about x, y, z                                             */
    const float    *op_arg;      /* [3 n_ops] offset | sin, cos, 0                                                     */
} orc_scene;

typedef struct orc_camera {      /* SimpleCamera members used by GetRay, 19 floats */
    float origin[3], top_left[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius;
} orc_camera;

typedef struct orc_hit {         /* Hitables.h:9-17 */
    float    t, pos[3], nrm[3], u, v;
    int32_t  mat;                /* material index, -1 = no hit */
    int32_t  prim;               /* winning primitive, -1 = no hit */
} orc_hit;

/* render flags (same values as MIRT_FLAG_*) */
enum {
    ORC_FLAG_ONE_SPP        = 1,   /* HomemadeRayTracer.cpp:114-121                         */
    ORC_FLAG_NORMAL_DISPLAY = 2,   /* HomemadeRayTracer.cpp:164-167                         */
    ORC_FLAG_ACCEL_LIST     = 4,   /* closest hit by a linear pass over all objects         */
    ORC_FLAG_ACCEL_ORDERED  = 32,  /* not in the reference: the library's near-first walk of the tree (rt_kernels.h
                                      closest_bvh2), restated for its work counters; same closest t            */
    ORC_FLAG_RECURSIVE      = 8    /* oracle only: evaluate emitted + atten*Sample() in the
                                      reference's recursive order instead of forward
                                      throughput order                                      */
};

/* RNG source kinds */
enum { ORC_RNG_PHILOX = 0, ORC_RNG_STREAM = 1, ORC_RNG_ENTROPY = 2 };

typedef struct orc_stats {
    uint64_t rays, nodes_visited, prims_tested, depth_cutoffs;
} orc_stats;

/* ---- Philox4x32-10 (the RNG *source*; replaces std::random_device, Randomizer.h:11,22) */
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Randomizer.h:9-26 mappings from one raw 32-bit draw */
float    orc_map_minmax(uint32_t raw, float mn, float mx);    /* [min,max)  :9-17  */
float    orc_map_minmax2(uint32_t raw, float mn, float mx);   /* [min,max]  :20-26 */
/* exhaustive self-check of the division-free form of raw/4294967295.0 used on the device */
uint64_t orc_check_pm1_short(uint32_t begin, uint32_t end);

/* SimpleTexture2D_TGAImage::Sample, SimpleTexture2D.cpp:130-147 */
void orc_tex_sample_batch(const orc_scene *sc, uint32_t tex, const float *uv, size_t n, float *rgb_out);

/* ---- primitive-level functions on raw arrays (test hooks) ---------------------------- */
/* AABB.cpp:5-21 */
void orc_aabb_hit_batch(const float *bmin, const float *bmax, size_t nbox,
                        const float *org, const float *dir, size_t nray,
                        float t_min, float t_max, uint8_t *out /* [nray*nbox] */);
/* one primitive through its whole instance chain (Hitables.cpp:163-173,241-264,13-54,93-115) */
void orc_prim_hit_batch(const orc_scene *sc, uint32_t prim, const float *org, const float *dir,
                        size_t nray, float t_min, float t_max, orc_hit *out);
/* closest hit in the world: BVH (SimpleObject.cpp:389-406) or list (flags & ORC_FLAG_ACCEL_LIST) */
void orc_world_hit_batch(const orc_scene *sc, const float *org, const float *dir, size_t nray,
                         float t_min, float t_max, uint32_t flags, orc_hit *out, orc_stats *st);
/* scatter with explicit raw draws: draws[i*draws_per_ray ...]; consumed[i] = raw draws used.
   out_alive: return value of Scatter; atten/dir/emitted as in Materials.cpp */
void orc_scatter_batch(const orc_scene *sc, const orc_hit *hits, const float *dir_in, size_t n,
                       const uint32_t *draws, uint32_t draws_per_ray,
                       uint8_t *out_alive, float *out_atten, float *out_org, float *out_dir,
                       float *out_emit, uint32_t *consumed);
/* SimpleCamera.cpp:55-60 with explicit raw draws (disk rejection) */
void orc_camera_rays(const orc_camera *cam, const float *uv, size_t n,
                     const uint32_t *draws, uint32_t draws_per_ray,
                     float *out_org, float *out_dir, uint32_t *consumed);

/* ---- host-side builders ---------------------------------------------------------------- */
/* SimpleCamera.cpp:32-53,139-191 */
void orc_camera_init(orc_camera *cam, const float look_from[3], const float look_at[3],
                     float fov_deg, float aperture, float aspect);
/* object bounding boxes (Hitables.cpp:56-60,117-128,152-161,175-180,182-239) -> bmin/bmax [3*n_objects] */
void orc_object_bounds(const orc_scene *sc, float *bmin, float *bmax);
/* SimpleObject.cpp:293-346 with the random axis taken from axis_draws (raw 32-bit draws,
   consumed in construction order); returns node count, fills arrays sized >= 2*n_objects */
int32_t orc_bvh_build_reference(const orc_scene *sc, const uint32_t *axis_draws, size_t n_draws,
                                float *node_min, float *node_max, int32_t *node_left,
                                int32_t *node_right, int32_t *root_out);

/* ---- the render loop ------------------------------------------------------------------- */
/* HomemadeRayTracer.cpp:90-190.  Adds samples [spp_begin, spp_begin+spp_count) of every pixel,
   in sample order, to sum_rgb (W*H*3 floats, caller zero-initialises).  rng_kind PHILOX is keyed
   by (seed; pixel, sample, stage, block); ENTROPY draws from the OS per call like
   std::random_device (Randomizer.h:11,22).  n_threads <= 0 -> omp default. */
void orc_trace(const orc_scene *sc, const orc_camera *cam, uint32_t width, uint32_t height,
               uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth, uint64_t seed,
               uint32_t flags, int rng_kind, int n_threads, float *sum_rgb, orc_stats *st);
/* HomemadeRayTracer.cpp:134-138 + OutputImage.cpp:62-88: mean (unless ONE_SPP), sqrt, clamp,
   (UINT32)(c*255.0f)&0xFF, A=255.  pixels_out (W*H*3, post-gamma floats) may be NULL. */
void orc_resolve(const float *sum_rgb, uint32_t width, uint32_t height, uint32_t spp_total,
                 uint32_t flags, float *pixels_out, uint8_t *rgba8_out);
/* PPMImageMaker.cpp:31-43 (ASCII P3, alpha dropped); returns 0 on success */
int  orc_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgba8);

int  orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
