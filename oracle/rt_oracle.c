/*
 * rt_oracle.c -- CPU restatement of HomemadeRayTracer's render loop.  TEST INFRASTRUCTURE ONLY.
 * See rt_oracle.h for the scope rules and the "parity unpinned" statement.
 *
 * Build: gcc -std=c11 -O2 -fopenmp -ffp-contract=off -fno-fast-math -fPIC -shared
 * Citations are relative to /root/reference/RayTracer/RayTracer/.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ===================================================================================== */
/* Vec3.h / Ray.h : thin wrappers over DirectXMath; the operation order below is the SSE2
 * path's (see header).                                                                   */
/* ===================================================================================== */
typedef struct { float x, y, z; } v3;
typedef struct { v3 org, dir; } ray_t;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline float v_get(v3 a, unsigned i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }      /* Vec3.h:22 */
static inline void v_set(v3 *a, unsigned i, float f) { if (i == 0) a->x = f; else if (i == 1) a->y = f; else a->z = f; } /* :23 */
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }                /* :55-58 */
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }                /* :60-63 */
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }                /* :65-68 */
static inline v3 v_scale(v3 a, float f) { return V(a.x * f, a.y * f, a.z * f); }                 /* :75-83 XMVectorScale */
static inline v3 v_divf(v3 a, float f) { return v_scale(a, 1.f / f); }                           /* :85-88  v / f == v * (1/f) */
static inline v3 v_neg(v3 a) { return V(0.0f - a.x, 0.0f - a.y, 0.0f - a.z); }                   /* :21 XMVectorNegate = 0 - v */
static inline float v_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }            /* :90-93 XMVector3Dot */
static inline float v_len2(v3 a) { return v_dot(a, a); }                                         /* :33 */
static inline float v_len(v3 a) { return sqrtf(v_dot(a, a)); }                                   /* :32 */
static inline v3 v_cross(v3 a, v3 b)                                                             /* :95-98 XMVector3Cross */
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v_normalize(v3 a)                                                               /* :100-103 XMVector3Normalize */
{
    float l2 = v_dot(a, a);
    float l = sqrtf(l2);
    if (l == 0.0f) return V(0.0f, 0.0f, 0.0f);               /* zero-length -> zero vector */
    if (isinf(l2)) { float q = NAN; return V(q, q, q); }     /* infinite length -> QNaN    */
    return V(a.x / l, a.y / l, a.z / l);                     /* true divide per lane       */
}
static inline v3 ray_at(const ray_t *r, float t) { return v_add(r->org, v_scale(r->dir, t)); }   /* Ray.h:14 */

/* Win32 min/max macros used by AABB.h:22-30 and AABB.cpp:15-16: plain compare-select */
#define W32_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define W32_MIN(a, b) (((a) < (b)) ? (a) : (b))

/* ===================================================================================== */
/* RNG source: Philox4x32-10 (Salmon et al., SC'11), replacing std::random_device.        */
/* ===================================================================================== */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Randomizer.h:9-17.  rd.max()+1 = 2^32; r in double; r*(max-min)+min in double; cast. */
float orc_map_minmax(uint32_t raw, float mn, float mx)
{
    double r = (double)raw / 4294967296.0;
    r = r * (double)(mx - mn) + (double)mn;
    return (float)r;
}
/* Randomizer.h:20-26.  divides by rd.max() = 2^32-1 -> [min,max] inclusive. */
float orc_map_minmax2(uint32_t raw, float mn, float mx)
{
    double r = (double)raw / 4294967295.0;
    r = r * (double)(mx - mn) + (double)mn;
    return (float)r;
}
/* The device maps a word to (-1,1) without the divide: (float)fma(raw * R, 2.0, -1.0) with R = 1/4294967295.0 correctly
 * rounded.  raw * R can be one ulp off the IEEE quotient, but that bit never reaches the float.  Returns the number of raw
 * values in [begin,end] (inclusive) for which the short form differs from orc_map_minmax2(raw, -1, 1), the reference's form. */
uint64_t orc_check_pm1_short(uint32_t begin, uint32_t end)
{
    const double R = 1.0 / 4294967295.0;
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t i = (int64_t)begin; i <= (int64_t)end; ++i) {
        const uint32_t raw = (uint32_t)i;
        const float want = orc_map_minmax2(raw, -1.0f, 1.0f);
        const float got = (float)fma((double)raw * R, 2.0, -1.0);
        uint32_t x, y; memcpy(&x, &want, 4); memcpy(&y, &got, 4);
        bad += (x != y);
    }
    return bad;
}

typedef struct {
    int            kind;
    /* philox */
    uint32_t       key[2], pixel, sample, stage, lanes, k, cache_block, cache[4];
    int            cache_ok;
    /* stream */
    const uint32_t *stream;
    uint32_t       stream_n, stream_pos;
} rng_t;

static void rng_stage(rng_t *g, uint32_t stage, uint32_t lanes, uint32_t first_k)
{
    g->stage = stage; g->lanes = lanes; g->k = first_k; g->cache_ok = 0;
}
/* one raw 32-bit draw == one `rd()` of Randomizer.h:14,23 */
static uint32_t rng_raw(rng_t *g)
{
    if (g->kind == ORC_RNG_PHILOX) {
        uint32_t block = g->k / g->lanes, lane = g->k % g->lanes;
        if (!g->cache_ok || g->cache_block != block) {
            uint32_t ctr[4] = { g->pixel, g->sample, g->stage, block };
            orc_philox4x32_10(ctr, g->key, g->cache);
            g->cache_block = block; g->cache_ok = 1;
        }
        g->k++;
        return g->cache[lane];
    }
    if (g->kind == ORC_RNG_STREAM) {
        uint32_t r = (g->stream_pos < g->stream_n) ? g->stream[g->stream_pos] : 0u;
        g->stream_pos++;
        return r;
    }
    /* ORC_RNG_ENTROPY: a fresh OS entropy read per draw, the closest Linux analogue of
     * constructing a std::random_device per call (Randomizer.h:11,22). */
    uint32_t r = 0;
    if (getrandom(&r, sizeof r, 0) != (ssize_t)sizeof r) r = (uint32_t)rand();
    return r;
}
static inline float rng_unorm(rng_t *g) { return orc_map_minmax(rng_raw(g), 0.0f, 1.0f); }       /* Randomizer.h:29-32 */

#define ORC_REJECT_CAP 64  /* the reference loops unboundedly; P(64 rejections) < 1e-20 */

/* Randomizer.h:40-48.  Philox layout: iteration i uses lanes 0..2 of block i (lanes=3). */
static v3 rng_in_unit_sphere(rng_t *g)
{
    v3 p = V(0, 0, 0);
    for (int it = 0; it < ORC_REJECT_CAP; ++it) {
        float x = orc_map_minmax2(rng_raw(g), -1.0f, 1.0f);
        float y = orc_map_minmax2(rng_raw(g), -1.0f, 1.0f);
        float z = orc_map_minmax2(rng_raw(g), -1.0f, 1.0f);
        p = V(x, y, z);
        if (!(v_len2(p) >= 1.0f)) break;
    }
    return p;
}
/* Randomizer.h:50-58 */
static v3 rng_in_unit_disk(rng_t *g)
{
    v3 p = V(0, 0, 0);
    for (int it = 0; it < ORC_REJECT_CAP; ++it) {
        float x = orc_map_minmax2(rng_raw(g), -1.0f, 1.0f);
        float y = orc_map_minmax2(rng_raw(g), -1.0f, 1.0f);
        p = V(x, y, 0.0f);
        if (!(v_len2(p) >= 1.0f)) break;
    }
    return p;
}

/* ===================================================================================== */
/* AABB.cpp:5-21                                                                          */
/* ===================================================================================== */
static int aabb_hit(v3 bmin, v3 bmax, const ray_t *r, float t_min, float t_max)
{
    for (unsigned a = 0; a < 3; ++a) {
        float invD = 1.0f / v_get(r->dir, a);
        float t0 = (v_get(bmin, a) - v_get(r->org, a)) * invD;
        float t1 = (v_get(bmax, a) - v_get(r->org, a)) * invD;
        if (invD < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        t_min = W32_MAX(t0, t_min);
        t_max = W32_MIN(t1, t_max);
        if (t_max <= t_min) return 0;
    }
    return 1;
}

/* ===================================================================================== */
/* Hitables.cpp                                                                           */
/* ===================================================================================== */
typedef struct { float t; v3 pos, nrm; float u, v; int mat; int prim; } hit_t;

/* atan2f / asinf of CalculateUV, pinned by definition.  The reference calls the MSVC CRT's (Hitables.cpp:70-71), whose bits are
 * not known here, and glibc's differ from the device library's in the last place.  Like Schlick's pow, the oracle and the
 * device therefore evaluate ONE stated sequence of IEEE float operations (rt_device.h det_atan2f / det_asinf: classic argument
 * reductions + single-precision minimax polynomials, error below 3 ulp): they agree bit for bit, and both stay closer to any
 * correct libm than two libms are to each other. */
static float det_atanf_pos(float x)
{
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    const float z = x * x;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return y + p;
}
static float det_atan2f(float y, float x)
{
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966f;
    if (x != x || y != y) return NAN;
    if (x == 0.0f) return y > 0.0f ? PIO2 : (y < 0.0f ? -PIO2 : 0.0f);
    if (y == 0.0f) return x < 0.0f ? PI : 0.0f;
    const float q = y / x;
    const float a = det_atanf_pos(fabsf(q));
    const float z = q < 0.0f ? -a : a;
    const float w = x < 0.0f ? (y < 0.0f ? -PI : PI) : 0.0f;
    return w + z;
}
static float det_asinf(float x)
{
    const float PIO2 = 1.5707963267948966f;
    const float a = fabsf(x);
    if (!(a <= 1.0f)) return NAN;
    if (a < 1.0e-4f) return x;
    float z, s;
    const int big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); s = sqrtf(z); } else { s = a; z = s * s; }
    float r = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    if (big) { r = r + r; r = PIO2 - r; }
    return x < 0.0f ? -r : r;
}

/* Hitables.cpp:62-78 (always executed by SphereHitable::Hit) */
static void sphere_uv(hit_t *rec, v3 center)
{
    v3 p = v_normalize(v_sub(rec->pos, center));
    float theta = det_atan2f(p.z, p.x);
    float phi = det_asinf(p.y);
    rec->u = 1.0f - (theta + (float)M_PI) / (2.0f * (float)M_PI);
    rec->v = (phi + (float)M_PI / 2.0f) / (float)M_PI;
}

/* Hitables.cpp:13-54 */
static int sphere_hit(v3 center, float radius, int mat, const ray_t *r, float t_min, float t_max, hit_t *rec)
{
    v3 oc = v_sub(r->org, center);
    float a = v_dot(r->dir, r->dir);
    float b = 2.0f * v_dot(r->dir, oc);
    float c = v_dot(oc, oc) - radius * radius;
    float disc = b * b - 4 * a * c;
    if (disc > 0) {
        float t = (-b - sqrtf(disc)) / (2.0f * a);
        if (t <= t_max && t >= t_min) {
            rec->t = t;
            rec->pos = ray_at(r, t);
            rec->nrm = v_divf(v_sub(rec->pos, center), radius);
            rec->mat = mat;
            sphere_uv(rec, center);
            return 1;
        }
        t = (-b + sqrtf(disc)) / (2.0f * a);
        if (t <= t_max && t >= t_min) {
            rec->t = t;
            rec->pos = ray_at(r, t);
            rec->nrm = v_divf(v_sub(rec->pos, center), radius);
            rec->mat = mat;
            sphere_uv(rec, center);
            return 1;
        }
    }
    return 0;
}

typedef struct { unsigned aAx, bAx, cAx; float a0, a1, b0, b1, c; int reverse; } rect_t;

/* Hitables.cpp:93-115 */
static int rect_hit(const rect_t *q, int mat, const ray_t *r, float t_min, float t_max, hit_t *rec)
{
    float t = (q->c - v_get(r->org, q->cAx)) / v_get(r->dir, q->cAx);
    if (t <= t_max && t >= t_min) {
        float a = v_get(r->org, q->aAx) + t * v_get(r->dir, q->aAx);
        float b = v_get(r->org, q->bAx) + t * v_get(r->dir, q->bAx);
        if (a >= q->a0 && a <= q->a1 && b >= q->b0 && b <= q->b1) {
            rec->u = (a - q->a0) / (q->a1 - q->a0);
            rec->v = (b - q->b0) / (q->b1 - q->b0);
            rec->t = t;
            rec->mat = mat;
            rec->pos = ray_at(r, t);
            v_set(&rec->nrm, q->aAx, 0.0f);
            v_set(&rec->nrm, q->bAx, 0.0f);
            v_set(&rec->nrm, q->cAx, q->reverse ? -1.0f : 1.0f);
            return 1;
        }
    }
    return 0;
}

/* RotatedInstance axis tables, Hitables.cpp:188-206 */
static void rot_axes(unsigned rotAxis, unsigned *a, unsigned *b, float *OP)
{
    switch (rotAxis) {
    case 0:  *a = 1; *b = 2; *OP = -1.0f; break;
    case 1:  *a = 0; *b = 2; *OP = 1.0f;  break;
    default: *a = 0; *b = 1; *OP = -1.0f; break;
    }
}
/* Hitables.cpp:244-250 : inverse-rotate the ray into the child's space */
static ray_t rot_ray_in(const ray_t *r, unsigned rotAxis, float s, float c)
{
    unsigned a, b; float OP; rot_axes(rotAxis, &a, &b, &OP);
    float opp = -OP;
    ray_t m = *r;
    v_set(&m.org, a, c * v_get(r->org, a) + opp * s * v_get(r->org, b));
    v_set(&m.org, b, -opp * s * v_get(r->org, a) + c * v_get(r->org, b));
    v_set(&m.dir, a, c * v_get(r->dir, a) + opp * s * v_get(r->dir, b));
    v_set(&m.dir, b, -opp * s * v_get(r->dir, a) + c * v_get(r->dir, b));
    return m;
}
/* Hitables.cpp:253-260 : forward-rotate position and normal of the hit */
static void rot_hit_out(hit_t *rec, unsigned rotAxis, float s, float c)
{
    unsigned a, b; float OP; rot_axes(rotAxis, &a, &b, &OP);
    v3 pos = rec->pos, nor = rec->nrm;
    v_set(&pos, a, c * v_get(rec->pos, a) + OP * s * v_get(rec->pos, b));
    v_set(&pos, b, -OP * s * v_get(rec->pos, a) + c * v_get(rec->pos, b));
    v_set(&nor, a, c * v_get(rec->nrm, a) + OP * s * v_get(rec->nrm, b));
    v_set(&nor, b, -OP * s * v_get(rec->nrm, a) + c * v_get(rec->nrm, b));
    rec->pos = pos; rec->nrm = nor;
}

static void unpack_rect(const orc_scene *sc, uint32_t p, rect_t *q, int *has_rot)
{
    uint32_t ax = sc->prim_axes[p];
    const float *g = sc->prim_geom + 5 * (size_t)p;
    q->aAx = ax & 3; q->bAx = (ax >> 2) & 3; q->cAx = (ax >> 4) & 3;
    q->reverse = (ax >> 6) & 1; *has_rot = (ax >> 7) & 1;
    q->a0 = g[0]; q->a1 = g[1]; q->b0 = g[2]; q->b1 = g[3]; q->c = g[4];
}

/* One primitive through its instance chain, exactly as the factories build it:
 *   sphere: TranslatedInstance(SphereHitable(0,r))                       SimpleObject.cpp:136
 *   rect  : TranslatedInstance(RotZ(RotY(RotX(AxisAlignedRectHitable)))) SimpleObject.cpp:173-181
 * Hit: Hitables.cpp:163-173 (translate) -> :241-264 (rotate, outermost = z) -> leaf.     */
/* A primitive in chain form (include/mirt.h): the leaf with the instances around it, outermost first.  This is the reference's
 * recursion written as a loop: every TranslatedInstance::Hit (Hitables.cpp:163-173) / RotatedInstance::Hit (:241-264) first
 * transforms the ray and calls its child, and on a hit transforms the record on the way back out -- so the ray passes the ops
 * outermost to innermost, the record innermost to outermost.  The sphere keeps its own centre (Hitables.cpp:20,37,65). */
static v3 ld3(const float *p);
static int chain_form(const orc_scene *sc, uint32_t p) { return sc->prim_chain && sc->prim_chain[p] != 0; }
static int prim_hit_chain(const orc_scene *sc, uint32_t p, const ray_t *r, float t_min, float t_max, hit_t *rec)
{
    const uint32_t first = sc->prim_chain[p] - 1u, n = sc->prim_nops[p];
    int mat = (int)sc->prim_mat[p];
    ray_t cur = *r;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t kind = sc->op_kind[first + k]; const float *a = sc->op_arg + 3 * (size_t)(first + k);
        if (kind == 0) { ray_t moved = { v_sub(cur.org, V(a[0], a[1], a[2])), cur.dir }; cur = moved; }     /* :166 */
        else cur = rot_ray_in(&cur, kind - 1u, a[0], a[1]);                                              /* :244-250 */
    }
    int hit;
    if (sc->prim_kind[p] == ORC_PRIM_SPHERE)
        hit = sphere_hit(ld3(sc->prim_center + 3 * (size_t)p), sc->prim_geom[5 * (size_t)p], mat, &cur, t_min, t_max, rec);
    else {
        rect_t q; int has_rot; unpack_rect(sc, p, &q, &has_rot);
        hit = rect_hit(&q, mat, &cur, t_min, t_max, rec);
    }
    if (hit) {
        for (uint32_t k = n; k-- > 0;) {
            const uint32_t kind = sc->op_kind[first + k]; const float *a = sc->op_arg + 3 * (size_t)(first + k);
            if (kind == 0) rec->pos = v_add(rec->pos, V(a[0], a[1], a[2]));                              /* :169 */
            else rot_hit_out(rec, kind - 1u, a[0], a[1]);                                                /* :253-260 */
        }
        rec->prim = (int)p;
    }
    return hit;
}

static int prim_hit(const orc_scene *sc, uint32_t p, const ray_t *r, float t_min, float t_max, hit_t *rec)
{
    if (chain_form(sc, p)) return prim_hit_chain(sc, p, r, t_min, t_max, rec);
    v3 off = V(sc->prim_off[3 * p], sc->prim_off[3 * p + 1], sc->prim_off[3 * p + 2]);
    ray_t moved = { v_sub(r->org, off), r->dir };                          /* :166 */
    int mat = (int)sc->prim_mat[p];
    int hit;
    if (sc->prim_kind[p] == ORC_PRIM_SPHERE) {
        hit = sphere_hit(V(0.0f, 0.0f, 0.0f), sc->prim_geom[5 * (size_t)p], mat, &moved, t_min, t_max, rec);
    } else {
        rect_t q; int has_rot; unpack_rect(sc, p, &q, &has_rot);
        if (has_rot) {
            const float *sn = sc->prim_rot + 6 * (size_t)p;                /* sinx,cosx,siny,cosy,sinz,cosz */
            ray_t rz = rot_ray_in(&moved, 2, sn[4], sn[5]);
            ray_t ry = rot_ray_in(&rz, 1, sn[2], sn[3]);
            ray_t rx = rot_ray_in(&ry, 0, sn[0], sn[1]);
            hit = rect_hit(&q, mat, &rx, t_min, t_max, rec);
            if (hit) {
                rot_hit_out(rec, 0, sn[0], sn[1]);
                rot_hit_out(rec, 1, sn[2], sn[3]);
                rot_hit_out(rec, 2, sn[4], sn[5]);
            }
        } else {
            hit = rect_hit(&q, mat, &moved, t_min, t_max, rec);
        }
    }
    if (hit) { rec->pos = v_add(rec->pos, off); rec->prim = (int)p; }      /* :169 */
    return hit;
}

/* Object::Hit (SimpleObject.cpp:73-83) over the object's hitable; an object of several prims
 * is a HitableCombo (Hitables.cpp:130-142): linear pass, t_max shrinks on every hit.  The
 * combo sits inside the shared transform chain, which is deterministic per prim, so applying
 * it per face gives the same bits. */
static int object_hit(const orc_scene *sc, uint32_t obj, const ray_t *r, float t_min, float *t_max,
                      hit_t *rec, orc_stats *st)
{
    int any = 0;
    uint32_t first = sc->obj_first[obj], cnt = sc->obj_count[obj];
    float tm = *t_max;
    for (uint32_t i = 0; i < cnt; ++i) {
        if (st) st->prims_tested++;
        if (prim_hit(sc, first + i, r, t_min, tm, rec)) { tm = rec->t; any = 1; }
    }
    if (any) *t_max = rec->t;                                              /* :78 */
    return any;
}

/* SimpleObjectBVHNode::Hit, SimpleObject.cpp:389-406.  t_max is a float& shared by the whole
 * traversal; the box test takes it by value (AABB.cpp:5). */
static int bvh_hit(const orc_scene *sc, int32_t child, const ray_t *r, float t_min, float *t_max,
                   hit_t *rec, orc_stats *st)
{
    if (child == ORC_CHILD_NONE) return 0;
    if (child < 0) return object_hit(sc, (uint32_t)~child, r, t_min, t_max, rec, st);
    int hitMe = 0;
    const float *mn = sc->node_min + 3 * (size_t)child, *mx = sc->node_max + 3 * (size_t)child;
    if (st) st->nodes_visited++;
    if (aabb_hit(V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2]), r, t_min, *t_max)) {
        hitMe |= bvh_hit(sc, sc->node_left[child], r, t_min, t_max, rec, st);
        hitMe |= bvh_hit(sc, sc->node_right[child], r, t_min, t_max, rec, st);
    }
    return hitMe;
}

static int bvh_hit_ordered(const orc_scene *sc, const ray_t *r, float t_min, float *t_max, hit_t *rec, orc_stats *st);

static int world_hit(const orc_scene *sc, const ray_t *r, float t_min, float t_max, uint32_t flags,
                     hit_t *rec, orc_stats *st)
{
    rec->mat = -1; rec->prim = -1;
    if (st) st->rays++;
    if (flags & ORC_FLAG_ACCEL_ORDERED) return bvh_hit_ordered(sc, r, t_min, &t_max, rec, st);
    if (flags & ORC_FLAG_ACCEL_LIST) {
        int any = 0;
        for (uint32_t o = 0; o < sc->n_objects; ++o) any |= object_hit(sc, o, r, t_min, &t_max, rec, st);
        return any;
    }
    return bvh_hit(sc, sc->root, r, t_min, &t_max, rec, st);
}

/* ===================================================================================== */
/* Optics.h                                                                               */
/* ===================================================================================== */
static v3 reflect(v3 v, v3 n) { return v_sub(v, v_scale(n, 2 * v_dot(v, n))); }                  /* Optics.h:8-12 */
static int refract(v3 v, v3 n, float ni_over_nt, v3 *refracted)                                  /* Optics.h:14-23 */
{
    float dt = v_dot(v, n);
    float disc = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    int ret = disc > 0;
    if (ret) *refracted = v_sub(v_scale(v_sub(v, v_scale(n, dt)), ni_over_nt), v_scale(n, sqrtf(disc)));
    return ret;
}
/* Optics.h:27-32.  `pow(float,int)`: MSVC 14.1 <cmath> forwards to powf(x,(float)y) and the sum
 * stays in float.  powf is not bit-specified; restated here as x^5 evaluated in double and
 * rounded once (within 1 ulp of any faithful powf). */
static float pow5f(float x) { double d = x, d2 = d * d, d4 = d2 * d2; return (float)(d4 * d); }
static float schlick(float cosine, float ref_idx)
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * pow5f(1 - cosine);
}

/* ===================================================================================== */
/* Materials.cpp : Scatter                                                                */
/* ===================================================================================== */
typedef struct { int alive; v3 atten; ray_t out; v3 emitted; } scat_t;

/* SimpleTexture2D_TGAImage::Sample, SimpleTexture2D.cpp:130-147: nearest texel, i = UINT32(u*W),
 * j = UINT32((1-v)*H - 0.001f), clamped to the image (the `< 0` tests are dead for UINT32, :135-136);
 * channel / 255.0f.  The float -> UINT32 conversion is MSVC x64's cvttss2si to 64 bits truncated to 32:
 * values in (-1, 0) give 0, NaN and out-of-range give 0x8000000000000000 -> 0. */
static uint32_t f2u32_msvc(float f)
{
    if (!(f > -9223372036854775808.0f && f < 9223372036854775808.0f)) return 0u;
    return (uint32_t)(uint64_t)(int64_t)f;
}
static v3 tex_sample(const orc_scene *sc, uint32_t tex, float u, float v)
{
    uint32_t W = sc->tex_width[tex], H = sc->tex_height[tex];
    uint32_t i = f2u32_msvc(u * (float)W);
    uint32_t j = f2u32_msvc((1.0f - v) * (float)H - 0.001f);
    if (i > W - 1) i = W - 1;
    if (j > H - 1) j = H - 1;
    const uint8_t *px = sc->tex_rgba8 + 4 * ((size_t)sc->tex_offset[tex] + (size_t)j * W + i);
    return V((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}
void orc_tex_sample_batch(const orc_scene *sc, uint32_t tex, const float *uv, size_t n, float *rgb_out)
{
    for (size_t k = 0; k < n; ++k) { v3 c = tex_sample(sc, tex, uv[2 * k], uv[2 * k + 1]); rgb_out[3 * k] = c.x; rgb_out[3 * k + 1] = c.y; rgb_out[3 * k + 2] = c.z; }
}
/* m_albedo->Sample(rec.m_u, rec.m_v), Materials.cpp:30,104 */
static v3 mat_albedo(const orc_scene *sc, int m, float u, float v)
{
    if (sc->n_textures && sc->mat_texture && sc->mat_texture[m] >= 0) return tex_sample(sc, (uint32_t)sc->mat_texture[m], u, v);
    return V(sc->mat_albedo[3 * m], sc->mat_albedo[3 * m + 1], sc->mat_albedo[3 * m + 2]);
}

static void scatter(const orc_scene *sc, const ray_t *r_in, const hit_t *rec, rng_t *g, scat_t *o)
{
    int m = rec->mat;
    o->emitted = V(0.0f, 0.0f, 0.0f);                                      /* HomemadeRayTracer.cpp:173 */
    o->atten = V(0.0f, 0.0f, 0.0f);
    o->out.org = rec->pos; o->out.dir = V(0.0f, 0.0f, 0.0f);
    switch (sc->mat_kind[m]) {
    case ORC_MAT_LAMBERTIAN: {                                             /* Materials.cpp:19-33 */
        v3 target = v_add(v_add(rec->pos, rec->nrm), rng_in_unit_sphere(g));
        o->out.org = rec->pos; o->out.dir = v_sub(target, rec->pos);
        o->atten = mat_albedo(sc, m, rec->u, rec->v);                      /* SimpleTexture2D.h:39, SimpleTexture2D.cpp:130 */
        o->alive = v_dot(r_in->dir, rec->nrm) < 0;
        break;
    }
    case ORC_MAT_METAL: {                                                  /* Materials.cpp:99-106 */
        v3 refl = reflect(v_normalize(r_in->dir), rec->nrm);
        o->out.org = rec->pos;
        o->out.dir = v_add(refl, v_scale(rng_in_unit_sphere(g), sc->mat_fuzz[m]));
        o->atten = mat_albedo(sc, m, rec->u, rec->v);
        o->alive = v_dot(o->out.dir, rec->nrm) > 0;
        break;
    }
    case ORC_MAT_DIELECTRIC: {                                             /* Materials.cpp:172-217 */
        float ior = sc->mat_ior[m];
        v3 uv = v_normalize(r_in->dir), outward, refracted = V(0.0f, 0.0f, 0.0f);
        float ni_over_nt, cosine, reflect_prob;
        o->atten = V(1.0f, 1.0f, 1.0f);
        if (v_dot(uv, rec->nrm) > 0) {
            outward = v_neg(rec->nrm);
            ni_over_nt = ior;
            cosine = ior * v_dot(uv, rec->nrm);
        } else {
            outward = rec->nrm;
            ni_over_nt = 1.0f / ior;
            cosine = -v_dot(uv, rec->nrm);
        }
        if (refract(uv, outward, ni_over_nt, &refracted)) reflect_prob = schlick(cosine, ior);
        else reflect_prob = 1.0f;
        if (rng_unorm(g) < reflect_prob) o->out.dir = reflect(uv, rec->nrm);
        else o->out.dir = refracted;
        o->out.org = rec->pos;
        o->alive = 1;
        break;
    }
    default:                                                               /* DiffuseLight, Materials.cpp:260-264 */
        o->emitted = V(sc->mat_emit[3 * m], sc->mat_emit[3 * m + 1], sc->mat_emit[3 * m + 2]);
        o->alive = 0;
        break;
    }
}

/* ===================================================================================== */
/* SimpleCamera                                                                           */
/* ===================================================================================== */
static v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* SimpleCamera.cpp:55-60 */
static ray_t camera_get_ray(const orc_camera *c, float u, float v, rng_t *g)
{
    v3 rd = v_scale(rng_in_unit_disk(g), c->lens_radius);
    v3 offset = v_add(v_scale(ld3(c->u), rd.x), v_scale(ld3(c->v), rd.y));
    v3 origin = ld3(c->origin);
    v3 d = v_add(ld3(c->top_left), v_scale(ld3(c->horizontal), u));
    d = v_add(d, v_scale(ld3(c->vertical), v));
    d = v_sub(d, origin);
    d = v_sub(d, offset);
    ray_t r = { v_add(origin, offset), d };
    return r;
}

/* SimpleCamera.cpp:32-53 (Initialize), :139-174 (Reset: origin/focus), :177-191 (InternalUpdate) */
void orc_camera_init(orc_camera *cam, const float look_from[3], const float look_at[3],
                     float fov_deg, float aperture, float aspect)
{
    v3 origin = ld3(look_from), focus = ld3(look_at), vup = V(0.0f, 1.0f, 0.0f);
    if (v_len(v_sub(focus, origin)) == 0.0f) focus = v_add(origin, V(0.0f, 0.0f, -1.0f));   /* :166-169 */
    float focusDist = v_len(v_sub(origin, focus));
    v3 w = v_normalize(v_sub(origin, focus));
    v3 u = v_normalize(v_cross(vup, w));
    v3 v = v_cross(w, u);
    const float theta = fov_deg * (float)M_PI / 180.0f;
    const float halfH = tanf(theta / 2.0f) * focusDist;
    const float halfW = halfH * aspect;
    v3 tl = v_sub(v_add(v_sub(origin, v_scale(u, halfW)), v_scale(v, halfH)), v_scale(w, focusDist));
    st3(cam->origin, origin); st3(cam->top_left, tl);
    st3(cam->horizontal, v_scale(u, 2.0f * halfW));
    st3(cam->vertical, v_scale(v, -2.0f * halfH));
    st3(cam->u, u); st3(cam->v, v);
    cam->lens_radius = aperture / 2.0f;                                    /* :48 */
}

/* ===================================================================================== */
/* Bounding boxes and the reference BVH build                                             */
/* ===================================================================================== */
typedef struct { v3 mn, mx; } box_t;
static box_t box_combine(box_t b0, box_t b1)                               /* AABB.h:19-34 */
{
    box_t r;
    r.mn = V(W32_MIN(b0.mn.x, b1.mn.x), W32_MIN(b0.mn.y, b1.mn.y), W32_MIN(b0.mn.z, b1.mn.z));
    r.mx = V(W32_MAX(b0.mx.x, b1.mx.x), W32_MAX(b0.mx.y, b1.mx.y), W32_MAX(b0.mx.z, b1.mx.z));
    return r;
}
static box_t rect_box(const rect_t *q)                                     /* Hitables.cpp:117-128 */
{
    box_t b; b.mn = V(0, 0, 0); b.mx = V(0, 0, 0);
    v_set(&b.mn, q->aAx, q->a0); v_set(&b.mx, q->aAx, q->a1);
    v_set(&b.mn, q->bAx, q->b0); v_set(&b.mx, q->bAx, q->b1);
    v_set(&b.mn, q->cAx, q->c - 0.0001f); v_set(&b.mx, q->cAx, q->c + 0.0001f);
    return b;
}
static box_t rot_box(box_t in, unsigned rotAxis, float s, float c)         /* Hitables.cpp:208-238 */
{
    unsigned aA, bA; float OP; rot_axes(rotAxis, &aA, &bA, &OP);
    unsigned cA = rotAxis;
    v3 mn = V(FLT_MAX, FLT_MAX, FLT_MAX), mx = V(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (unsigned i = 0; i < 2; ++i)
        for (unsigned j = 0; j < 2; ++j)
            for (unsigned k = 0; k < 2; ++k) {
                float a = i * v_get(in.mx, aA) + (1 - i) * v_get(in.mn, aA);
                float b = j * v_get(in.mx, bA) + (1 - j) * v_get(in.mn, bA);
                float cc = k * v_get(in.mx, cA) + (1 - k) * v_get(in.mn, cA);
                float newa = c * a + OP * s * b;
                float newb = -OP * s * a + c * b;
                v3 tester = V(0, 0, 0);
                v_set(&tester, aA, newa); v_set(&tester, bA, newb); v_set(&tester, cA, cc);
                for (unsigned ax = 0; ax < 3; ++ax) {
                    if (v_get(tester, ax) > v_get(mx, ax)) v_set(&mx, ax, v_get(tester, ax));
                    if (v_get(tester, ax) < v_get(mn, ax)) v_set(&mn, ax, v_get(tester, ax));
                }
            }
    box_t r = { mn, mx };
    return r;
}
/* A chain-form primitive's box: the leaf's, carried through its instances from the innermost out (Hitables.cpp:175-180, :208-238).
 * For a sphere this is the box AROUND ITS CENTRE -- the reference's SphereHitable::BoundingBox (Hitables.cpp:56-60) forgets the
 * centre, which is harmless for its factories (centre 0) and wrong for any other sphere; the boxes here are the library's own
 * (its builder and the object boxes of the near-first walk), which must contain what they guard. */
static box_t prim_box_chain(const orc_scene *sc, uint32_t p)
{
    box_t b;
    if (sc->prim_kind[p] == ORC_PRIM_SPHERE) {
        float r = sc->prim_geom[5 * (size_t)p]; v3 c = ld3(sc->prim_center + 3 * (size_t)p), rv = V(r, r, r);
        b.mn = v_sub(c, rv); b.mx = v_add(c, rv);
    } else { rect_t q; int hr; unpack_rect(sc, p, &q, &hr); b = rect_box(&q); }
    const uint32_t first = sc->prim_chain[p] - 1u, n = sc->prim_nops[p];
    for (uint32_t k = n; k-- > 0;) {
        const uint32_t kind = sc->op_kind[first + k]; const float *a = sc->op_arg + 3 * (size_t)(first + k);
        if (kind == 0) { v3 off = V(a[0], a[1], a[2]); b.mn = v_add(b.mn, off); b.mx = v_add(b.mx, off); }
        else b = rot_box(b, kind - 1u, a[0], a[1]);
    }
    return b;
}
static box_t object_box_factory(const orc_scene *sc, uint32_t first, uint32_t cnt)
{
    v3 off = ld3(sc->prim_off + 3 * (size_t)first);
    box_t b;
    if (sc->prim_kind[first] == ORC_PRIM_SPHERE) {                         /* Hitables.cpp:56-60 */
        float r = sc->prim_geom[5 * (size_t)first];
        v3 rv = V(r, r, r);
        b.mn = v_neg(rv); b.mx = rv;
    } else {
        rect_t q; int has_rot, hr;
        unpack_rect(sc, first, &q, &has_rot);
        b = rect_box(&q);
        for (uint32_t i = 1; i < cnt; ++i) {                               /* HitableCombo, Hitables.cpp:152-161 */
            rect_t qi; unpack_rect(sc, first + i, &qi, &hr);
            b = box_combine(rect_box(&qi), b);
        }
        if (has_rot) {
            const float *sn = sc->prim_rot + 6 * (size_t)first;
            b = rot_box(b, 0, sn[0], sn[1]);
            b = rot_box(b, 1, sn[2], sn[3]);
            b = rot_box(b, 2, sn[4], sn[5]);
        }
    }
    b.mn = v_add(b.mn, off); b.mx = v_add(b.mx, off);                      /* Hitables.cpp:175-180 */
    return b;
}
/* An object of factory-form primitives is one of the factories' shapes (a sphere, a rect, a cube's six faces inside ONE chain):
 * its box is built the way the reference builds it, combo first, then the chain.  An object with a chain-form primitive is a
 * general combo: the union, in order, of its primitives' boxes (Hitables.cpp:152-161). */
static box_t object_box(const orc_scene *sc, uint32_t obj)
{
    uint32_t first = sc->obj_first[obj], cnt = sc->obj_count[obj];
    int any_chain = 0;
    for (uint32_t i = 0; i < cnt; ++i) any_chain |= chain_form(sc, first + i);
    if (!any_chain) return object_box_factory(sc, first, cnt);
    box_t b = chain_form(sc, first) ? prim_box_chain(sc, first) : object_box_factory(sc, first, 1);
    for (uint32_t i = 1; i < cnt; ++i)
        b = box_combine(chain_form(sc, first + i) ? prim_box_chain(sc, first + i) : object_box_factory(sc, first + i, 1), b);
    return b;
}
/* NOT in the reference: the near-first walk the library runs under MIRT_FLAG_ACCEL_ORDERED / _AUTO (rt_kernels.h
 * closest_bvh2), restated on the host for its work counters (bench.py prices the BVH workloads with nodes_visited =
 * boxes tested and prims_tested of THIS walk, not of the reference's left-first one, which tests several times more) and as a
 * second opinion on its hits.  Each child's box is tested in the parent with AABB::Hit's arithmetic (AABB.cpp:5-21, 1/dir
 * hoisted) against the current t_max; the child entered nearer goes first.  A node's box culls on t_max <= t_min like
 * AABB.cpp:17; an object's box -- the reference never box-tests an object -- only on t_max < t_min.  The root's own box is
 * not tested.  Object boxes come from g_ordered_obox, set by the entry points below for the duration of a call. */
static const box_t *g_ordered_obox;
static int slab_ordered(box_t b, const ray_t *r, v3 invd, float t_min, float t_max, int is_object, float *t_near)
{
    for (unsigned a = 0; a < 3; ++a) {
        float id = v_get(invd, a);
        float t0 = (v_get(b.mn, a) - v_get(r->org, a)) * id;
        float t1 = (v_get(b.mx, a) - v_get(r->org, a)) * id;
        if (id < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        t_min = W32_MAX(t0, t_min);
        t_max = W32_MIN(t1, t_max);
    }
    *t_near = t_min;
    return is_object ? !(t_max < t_min) : !(t_max <= t_min);
}
static box_t ordered_child_box(const orc_scene *sc, int32_t c)
{
    if (c < 0) return g_ordered_obox[~c];
    box_t b = { ld3(sc->node_min + 3 * (size_t)c), ld3(sc->node_max + 3 * (size_t)c) };
    return b;
}
static int bvh_hit_ordered(const orc_scene *sc, const ray_t *r, float t_min, float *t_max, hit_t *rec, orc_stats *st)
{
    int32_t stack[128]; int sp = 0, any = 0;
    int32_t cur = sc->root;
    const v3 invd = V(1.0f / r->dir.x, 1.0f / r->dir.y, 1.0f / r->dir.z);
    for (;;) {
        if (cur >= 0) {
            const int32_t l = sc->node_left[cur], rr = sc->node_right[cur];
            float tl = 0.0f, tr = 0.0f;
            int hl = 0, hr = 0;
            if (l != ORC_CHILD_NONE) { if (st) st->nodes_visited++; hl = slab_ordered(ordered_child_box(sc, l), r, invd, t_min, *t_max, l < 0, &tl); }
            if (rr != ORC_CHILD_NONE) { if (st) st->nodes_visited++; hr = slab_ordered(ordered_child_box(sc, rr), r, invd, t_min, *t_max, rr < 0, &tr); }
            if (hl && hr) {
                const int left_first = tl <= tr;
                if (sp < 128) stack[sp++] = left_first ? rr : l;
                cur = left_first ? l : rr;
                continue;
            }
            if (hl) { cur = l; continue; }
            if (hr) { cur = rr; continue; }
        } else if (cur != ORC_CHILD_NONE) {
            any |= object_hit(sc, (uint32_t)~cur, r, t_min, t_max, rec, st);
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return any;
}
static box_t *ordered_begin(const orc_scene *sc, uint32_t flags)
{
    if (!(flags & ORC_FLAG_ACCEL_ORDERED)) return NULL;
    box_t *b = (box_t *)malloc((sc->n_objects ? sc->n_objects : 1) * sizeof *b);
    for (uint32_t o = 0; b && o < sc->n_objects; ++o) b[o] = object_box(sc, o);
    g_ordered_obox = b;
    return b;
}
static void ordered_end(box_t *b) { if (b) { g_ordered_obox = NULL; free(b); } }

void orc_object_bounds(const orc_scene *sc, float *bmin, float *bmax)
{
    for (uint32_t o = 0; o < sc->n_objects; ++o) {
        box_t b = object_box(sc, o);
        st3(bmin + 3 * (size_t)o, b.mn); st3(bmax + 3 * (size_t)o, b.mx);
    }
}

typedef struct {
    const orc_scene *sc; const box_t *obox; const uint32_t *draws; size_t n_draws, draw_pos;
    float *nmin, *nmax; int32_t *nl, *nr; int32_t n_nodes;
} bvhb_t;

/* stable merge sort, descending by translation[axis] (std::sort's order of ties is unspecified) */
static void sort_desc(uint32_t *idx, uint32_t *tmp, size_t n, const orc_scene *sc, unsigned axis)
{
    if (n < 2) return;
    size_t h = n / 2;
    sort_desc(idx, tmp, h, sc, axis); sort_desc(idx + h, tmp, n - h, sc, axis);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        float a = sc->prim_off[3 * (size_t)sc->obj_first[idx[i]] + axis];
        float b = sc->prim_off[3 * (size_t)sc->obj_first[idx[j]] + axis];
        /* comparator(a,b) = b.t < a.t  -> a sorts before b when a is greater */
        if (a < b) tmp[k++] = idx[j++]; else tmp[k++] = idx[i++];
    }
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, n * sizeof *idx);
}
static box_t child_box(const bvhb_t *b, int32_t c)
{
    if (c < 0) return b->obox[~c];
    box_t r = { ld3(b->nmin + 3 * (size_t)c), ld3(b->nmax + 3 * (size_t)c) };
    return r;
}
/* SimpleObject.cpp:293-346 */
static int32_t bvh_build_rec(bvhb_t *b, uint32_t *objs, uint32_t *tmp, size_t n)
{
    uint32_t raw = (b->draw_pos < b->n_draws) ? b->draws[b->draw_pos] : 0u;
    b->draw_pos++;
    unsigned axis = (unsigned)(uint32_t)(3 * orc_map_minmax(raw, 0.0f, 1.0f));   /* :298 */
    if (axis > 2) axis = 2;                                                       /* else-branch :303 */
    sort_desc(objs, tmp, n, b->sc, axis);
    int32_t me = b->n_nodes++;
    int32_t l, r;
    if (n == 1) { l = ~(int32_t)objs[0]; r = ORC_CHILD_NONE; }
    else if (n == 2) { l = ~(int32_t)objs[0]; r = ~(int32_t)objs[1]; }
    else {
        size_t half = (size_t)(n * 0.5f);                                          /* :325 */
        l = bvh_build_rec(b, objs, tmp, half);
        r = bvh_build_rec(b, objs + half, tmp + half, n - half);
    }
    box_t bb = (r == ORC_CHILD_NONE) ? child_box(b, l) : box_combine(child_box(b, l), child_box(b, r));
    st3(b->nmin + 3 * (size_t)me, bb.mn); st3(b->nmax + 3 * (size_t)me, bb.mx);
    b->nl[me] = l; b->nr[me] = r;
    return me;
}
int32_t orc_bvh_build_reference(const orc_scene *sc, const uint32_t *axis_draws, size_t n_draws,
                                float *node_min, float *node_max, int32_t *node_left,
                                int32_t *node_right, int32_t *root_out)
{
    uint32_t n = sc->n_objects;
    if (n == 0) { *root_out = ORC_CHILD_NONE; return 0; }
    box_t *obox = (box_t *)malloc(sizeof(box_t) * n);
    uint32_t *objs = (uint32_t *)malloc(sizeof(uint32_t) * n), *tmp = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (uint32_t o = 0; o < n; ++o) { obox[o] = object_box(sc, o); objs[o] = o; }
    bvhb_t b = { sc, obox, axis_draws, n_draws, 0, node_min, node_max, node_left, node_right, 0 };
    *root_out = bvh_build_rec(&b, objs, tmp, n);
    free(obox); free(objs); free(tmp);
    return b.n_nodes;
}

/* ===================================================================================== */
/* batch test hooks                                                                       */
/* ===================================================================================== */
static void hit_store(orc_hit *o, const hit_t *h, int hit)
{
    if (!hit) { memset(o, 0, sizeof *o); o->mat = -1; o->prim = -1; return; }
    o->t = h->t; st3(o->pos, h->pos); st3(o->nrm, h->nrm); o->u = h->u; o->v = h->v;
    o->mat = h->mat; o->prim = h->prim;
}
static void hit_load(hit_t *h, const orc_hit *o)
{
    h->t = o->t; h->pos = ld3(o->pos); h->nrm = ld3(o->nrm); h->u = o->u; h->v = o->v;
    h->mat = o->mat; h->prim = o->prim;
}

void orc_aabb_hit_batch(const float *bmin, const float *bmax, size_t nbox,
                        const float *org, const float *dir, size_t nray,
                        float t_min, float t_max, uint8_t *out)
{
    for (size_t i = 0; i < nray; ++i) {
        ray_t r = { ld3(org + 3 * i), ld3(dir + 3 * i) };
        for (size_t b = 0; b < nbox; ++b)
            out[i * nbox + b] = (uint8_t)aabb_hit(ld3(bmin + 3 * b), ld3(bmax + 3 * b), &r, t_min, t_max);
    }
}

void orc_prim_hit_batch(const orc_scene *sc, uint32_t prim, const float *org, const float *dir,
                        size_t nray, float t_min, float t_max, orc_hit *out)
{
    for (size_t i = 0; i < nray; ++i) {
        ray_t r = { ld3(org + 3 * i), ld3(dir + 3 * i) };
        hit_t h; memset(&h, 0, sizeof h);
        int hit = prim_hit(sc, prim, &r, t_min, t_max, &h);
        hit_store(out + i, &h, hit);
    }
}

void orc_world_hit_batch(const orc_scene *sc, const float *org, const float *dir, size_t nray,
                         float t_min, float t_max, uint32_t flags, orc_hit *out, orc_stats *st)
{
    orc_stats s; memset(&s, 0, sizeof s);
    box_t *ob = ordered_begin(sc, flags);
    for (size_t i = 0; i < nray; ++i) {
        ray_t r = { ld3(org + 3 * i), ld3(dir + 3 * i) };
        hit_t h; memset(&h, 0, sizeof h);
        int hit = world_hit(sc, &r, t_min, t_max, flags, &h, &s);
        hit_store(out + i, &h, hit);
    }
    ordered_end(ob);
    if (st) *st = s;
}

void orc_scatter_batch(const orc_scene *sc, const orc_hit *hits, const float *dir_in, size_t n,
                       const uint32_t *draws, uint32_t draws_per_ray,
                       uint8_t *out_alive, float *out_atten, float *out_org, float *out_dir,
                       float *out_emit, uint32_t *consumed)
{
    for (size_t i = 0; i < n; ++i) {
        rng_t g; memset(&g, 0, sizeof g);
        g.kind = ORC_RNG_STREAM; g.stream = draws + i * (size_t)draws_per_ray; g.stream_n = draws_per_ray;
        hit_t h; hit_load(&h, hits + i);
        ray_t rin = { V(0.0f, 0.0f, 0.0f), ld3(dir_in + 3 * i) };
        scat_t o; memset(&o, 0, sizeof o);
        scatter(sc, &rin, &h, &g, &o);
        out_alive[i] = (uint8_t)o.alive;
        st3(out_atten + 3 * i, o.atten); st3(out_org + 3 * i, o.out.org); st3(out_dir + 3 * i, o.out.dir);
        st3(out_emit + 3 * i, o.emitted);
        if (consumed) consumed[i] = g.stream_pos;
    }
}

void orc_camera_rays(const orc_camera *cam, const float *uv, size_t n,
                     const uint32_t *draws, uint32_t draws_per_ray,
                     float *out_org, float *out_dir, uint32_t *consumed)
{
    for (size_t i = 0; i < n; ++i) {
        rng_t g; memset(&g, 0, sizeof g);
        g.kind = ORC_RNG_STREAM; g.stream = draws + i * (size_t)draws_per_ray; g.stream_n = draws_per_ray;
        ray_t r = camera_get_ray(cam, uv[2 * i], uv[2 * i + 1], &g);
        st3(out_org + 3 * i, r.org); st3(out_dir + 3 * i, r.dir);
        if (consumed) consumed[i] = g.stream_pos;
    }
}

/* ===================================================================================== */
/* HomemadeRayTracer::Sample / TraceRay                                                   */
/* ===================================================================================== */
typedef struct {
    const orc_scene *sc; uint32_t flags, max_depth; rng_t *g; orc_stats *st;
} sctx_t;

/* HomemadeRayTracer.cpp:156-190, the reference's recursion, verbatim order of operations */
static v3 sample_recursive(const sctx_t *c, const ray_t *r, uint32_t depth)
{
    v3 col;
    hit_t rec; memset(&rec, 0, sizeof rec);
    float nearest = 0.001f, closestSoFar = FLT_MAX;
    if (world_hit(c->sc, r, nearest, closestSoFar, c->flags, &rec, c->st)) {
        if (c->flags & ORC_FLAG_NORMAL_DISPLAY) {
            col = v_scale(V(rec.nrm.x + 1.0f, rec.nrm.y + 1.0f, rec.nrm.z + 1.0f), 0.5f);
        } else {
            scat_t s; s.emitted = V(0.0f, 0.0f, 0.0f); s.alive = 0;
            if (depth < c->max_depth && rec.mat >= 0) {
                rng_stage(c->g, 1 + depth, 3, 0);
                scatter(c->sc, r, &rec, c->g, &s);
            } else if (c->st) c->st->depth_cutoffs++;
            if (s.alive) col = v_add(s.emitted, v_mul(s.atten, sample_recursive(c, &s.out, depth + 1)));
            else col = s.emitted;
        }
    } else {
        col = ld3(c->sc->ambient);                                         /* :186 */
    }
    return col;
}

/* Same estimator in forward (throughput) order, the order a wavefront renderer evaluates:
 * L = ((a0*a1)*...*a_{n-1}) * E.  Differs from the recursion only by float re-association of
 * the attenuation product (emitted is non-zero only at the terminal vertex). */
static v3 sample_forward(const sctx_t *c, const ray_t *r0)
{
    ray_t r = *r0;
    v3 thr = V(1.0f, 1.0f, 1.0f);
    for (uint32_t depth = 0;; ++depth) {
        hit_t rec; memset(&rec, 0, sizeof rec);
        if (!world_hit(c->sc, &r, 0.001f, FLT_MAX, c->flags, &rec, c->st)) return v_mul(thr, ld3(c->sc->ambient));
        if (c->flags & ORC_FLAG_NORMAL_DISPLAY)
            return v_scale(V(rec.nrm.x + 1.0f, rec.nrm.y + 1.0f, rec.nrm.z + 1.0f), 0.5f);
        if (!(depth < c->max_depth && rec.mat >= 0)) { if (c->st) c->st->depth_cutoffs++; return V(0.0f, 0.0f, 0.0f); }
        scat_t s;
        rng_stage(c->g, 1 + depth, 3, 0);
        scatter(c->sc, &r, &rec, c->g, &s);
        if (!s.alive) return v_mul(thr, s.emitted);
        thr = v_mul(thr, s.atten);
        r = s.out;
    }
}

void orc_trace(const orc_scene *sc, const orc_camera *cam, uint32_t width, uint32_t height,
               uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth, uint64_t seed,
               uint32_t flags, int rng_kind, int n_threads, float *sum_rgb, orc_stats *st)
{
    orc_stats total; memset(&total, 0, sizeof total);
    box_t *ordered_boxes = ordered_begin(sc, flags);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    /* HomemadeRayTracer.cpp:107-110: OpenMP over rows */
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t j = 0; j < (int32_t)height; ++j) {
        orc_stats ls; memset(&ls, 0, sizeof ls);
        rng_t g; memset(&g, 0, sizeof g);
        g.kind = rng_kind; g.key[0] = (uint32_t)seed; g.key[1] = (uint32_t)(seed >> 32);
        sctx_t c = { sc, flags, max_depth, &g, &ls };
        for (uint32_t i = 0; i < width; ++i) {
            size_t pix = (size_t)j * width + i;
            g.pixel = (uint32_t)pix;
            v3 col;
            if (flags & ORC_FLAG_ONE_SPP) {                                 /* :114-121 */
                float u = (float)i / (float)width;
                float v = (float)j / (float)height;
                g.sample = 0; rng_stage(&g, 0, 4, 2);
                ray_t r = camera_get_ray(cam, u, v, &g);
                col = (flags & ORC_FLAG_RECURSIVE) ? sample_recursive(&c, &r, 0) : sample_forward(&c, &r);
            } else {                                                        /* :125-133 */
                col = ld3(sum_rgb + 3 * pix);
                for (uint32_t s = spp_begin; s < spp_begin + spp_count; ++s) {
                    g.sample = s; rng_stage(&g, 0, 4, 0);
                    float u = ((float)i + rng_unorm(&g)) / (float)width;
                    float v = ((float)j + rng_unorm(&g)) / (float)height;
                    ray_t r = camera_get_ray(cam, u, v, &g);
                    col = v_add(col, (flags & ORC_FLAG_RECURSIVE) ? sample_recursive(&c, &r, 0) : sample_forward(&c, &r));
                }
            }
            st3(sum_rgb + 3 * pix, col);
        }
#pragma omp critical
        {
            total.rays += ls.rays; total.nodes_visited += ls.nodes_visited;
            total.prims_tested += ls.prims_tested; total.depth_cutoffs += ls.depth_cutoffs;
        }
    }
    ordered_end(ordered_boxes);
    if (st) *st = total;
}

/* XMVectorClamp lane: _mm_min_ps(hi, _mm_max_ps(lo, v)); maxps(a,b) = a > b ? a : b */
static inline float clamp01(float v)
{
    float m = (0.0f > v) ? 0.0f : v;
    return (1.0f < m) ? 1.0f : m;
}
static inline uint8_t to_byte(float c)                                     /* OutputImage.cpp:80-82 */
{
    float s = c * 255.0f;
    if (!(s == s)) return 0;         /* NaN: cvttss2si -> 0x80000000, & 0xFF = 0 */
    return (uint8_t)((uint32_t)s & 0xFF);
}

void orc_resolve(const float *sum_rgb, uint32_t width, uint32_t height, uint32_t spp_total,
                 uint32_t flags, float *pixels_out, uint8_t *rgba8_out)
{
    size_t n = (size_t)width * height;
    for (size_t p = 0; p < n; ++p) {
        v3 col = ld3(sum_rgb + 3 * p);
        if (!(flags & ORC_FLAG_ONE_SPP)) col = v_divf(col, (float)spp_total);           /* HomemadeRayTracer.cpp:134 */
        col = V(sqrtf(col.x), sqrtf(col.y), sqrtf(col.z));                              /* :138 */
        if (pixels_out) st3(pixels_out + 3 * p, col);
        if (rgba8_out) {                                                                /* OutputImage.cpp:76-83 */
            rgba8_out[4 * p + 0] = to_byte(clamp01(col.x));
            rgba8_out[4 * p + 1] = to_byte(clamp01(col.y));
            rgba8_out[4 * p + 2] = to_byte(clamp01(col.z));
            rgba8_out[4 * p + 3] = 255;
        }
    }
}

/* PPMImageMaker.cpp:31-43 */
int orc_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgba8)
{
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "P3\n%u %u\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * height; ++i)
        fprintf(f, "%u %u %u\n", (unsigned)rgba8[4 * i], (unsigned)rgba8[4 * i + 1], (unsigned)rgba8[4 * i + 2]);
    return fclose(f);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
