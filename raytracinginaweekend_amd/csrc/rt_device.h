// rt_device.h -- gfx950 device functions of the path tracer: float3 math in the
// reference's operation order, Philox4x32-10, the Randomizer mappings, intersection and scatter.
//
// Everything here is compiled with -ffp-contract=off: the reference is MSVC /fp:precise on the
// DirectXMath SSE2 path (no FMA), and parity with the CPU oracle is bit-exact only if no
// multiply-add is contracted.  Divisions and square roots are IEEE correctly rounded (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt).  Explicit fma() calls are used only where
// the result is proven equal to the plain form (map_pm1, checked on all 2^32 words; div_by and the lean square root, which
// run the hardware's own sequences).
//
// File:line citations are relative to /root/reference/RayTracer/RayTracer/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mirt.h"

namespace mirt {

// MIRT_TEXP: ablation bits for timing experiments on k_trace (tools/ablate_trace.sh, results in
// profiles/r02_trace_ablation.txt); never set in the product, results are wrong by design.  1: metal/dielectric shaded as
// Lambertian (upper bound of what regrouping by material can save), 2: hardware sqrt approximation instead of the
// correctly rounded expansion, 4: no rejection rounds after the first candidate, 8: no radiance store
#ifndef MIRT_TEXP
#define MIRT_TEXP 0
#endif
// The correctly rounded square root of an operand in [2^-60, 2^60]: the hardware's v_sqrt_f32 (within 1 ulp) and the choice
// between its neighbours by the sign of the two residuals x - s*(s -+ 1ulp) -- the core of the compiler's own 17-instruction
// expansion of sqrtf, without the seven instructions that rescale denormal operands and pass zero / infinity through.
__device__ __forceinline__ float sqrt_in_range(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}
// sqrtf for a wave: the lean form when every active lane's operand is in its range (which the squared lengths and
// discriminants of this path are, all but never), the compiler's general expansion for the whole wave otherwise.  Same bits.
__device__ __forceinline__ float sqrt_wave(float x)
{
    const bool in_range = (x >= 0x1p-60f) & (x <= 0x1p60f);                 // false for zero, denormals, negatives, infinity and NaN
    if (__builtin_amdgcn_ballot_w64(!in_range) == 0ull) return sqrt_in_range(x);
    return __builtin_sqrtf(x);
}
#define MIRT_SQRT(x) ((MIRT_TEXP & 2) ? __builtin_amdgcn_sqrtf(x) : sqrt_wave(x))

// ----------------------------------------------------------------------------- Vec3.h / Ray.h
struct f3 { float x, y, z; };
struct Ray { f3 o, d; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }   // Vec3.h:55-58
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }   // Vec3.h:60-63
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }   // Vec3.h:65-68
__device__ __forceinline__ f3 scale(f3 a, float f) { return mk(a.x * f, a.y * f, a.z * f); }          // Vec3.h:75-83
__device__ __forceinline__ f3 neg(f3 a) { return mk(0.0f - a.x, 0.0f - a.y, 0.0f - a.z); }            // Vec3.h:21 (0 - v)
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }      // Vec3.h:90-93
__device__ __forceinline__ float get(f3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__device__ __forceinline__ void set(f3 &a, uint32_t i, float f) { if (i == 0) a.x = f; else if (i == 1) a.y = f; else a.z = f; }
// ----------------------------------------------------------------------------- n / d for a divisor known before the launch
// The image width and the pixels of a batch divide every path id (sample = id / npix, row = pixel / width); the compiler's
// expansion of a 32-bit division by a run-time value is ~20 instructions.  With the divisor's 33-bit reciprocal made once on
// the host (the round-up method of Granlund & Montgomery, "add" form valid for every 32-bit n) it is a multiply-high, a
// subtract, two shifts and an add.  d == 1 is flagged (its reciprocal does not fit).
struct FastDiv { uint32_t m, sh, one, d; };
__host__ __device__ inline FastDiv fastdiv_make(uint32_t d)
{
    FastDiv f; f.d = d; f.one = d <= 1u ? 1u : 0u; f.m = 0u; f.sh = 0u;
    if (d <= 1u) return f;
    uint32_t l = 31u; while (!((d >> l) & 1u)) --l;                       // floor(log2 d)
    if ((d & (d - 1u)) == 0u) { f.m = 0u; f.sh = l - 1u; return f; }      // power of two: ((n - 0) >> 1) >> (l - 1)
    const unsigned long long two_l = 1ull << (32u + l);
    unsigned long long m = two_l / d; const unsigned long long rem = two_l - m * d;
    m = 2ull * m + ((2ull * rem >= d) ? 1ull : 0ull);                     // floor(2^(33+l) / d), low 32 bits kept below
    f.m = (uint32_t)(m + 1ull); f.sh = l;
    return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv &f)
{
    const uint32_t q = __umulhi(n, f.m);
    const uint32_t t = ((n - q) >> 1) + q;
    return f.one ? n : (t >> f.sh);                                      // f is wave-uniform: a scalar select
}

// ----------------------------------------------------------------------------- many IEEE quotients, one denominator
// x / den for several x (the rects of one orientation all divide by the same ray-direction component).  This is the
// hardware's own f32 division sequence (v_div_scale x2, v_rcp, 2 fma on the reciprocal, mul, 3 fma, v_div_fmas,
// v_div_fixup) with the part that depends on the denominator alone hoisted out: when neither operand needs
// v_div_scale's rescaling the scaled operands are the operands themselves, v_div_fmas is a plain fma and v_div_fixup
// passes its input through, so the five remaining operations give the very bits of `x / den`.
// Promise (tests/test_gpu_parity.py::test_shared_denominator_division checks it against `/` on 2^26 pairs and the
// edges): for |den| in [2^-20, 2^20] and |x| in [2^-40, 2^40] or x == +0 the result is x / den bit for bit; for
// smaller |x| (and x == -0) it is only guaranteed to stay below 2^-19 in magnitude, like the true quotient.
struct DivBy { float den, r1; };
__device__ __forceinline__ DivBy div_prepare(float den)
{
    DivBy q; q.den = den;
    const float r0 = __builtin_amdgcn_rcpf(den);
    q.r1 = __builtin_fmaf(__builtin_fmaf(-den, r0, 1.0f), r0, r0);
    return q;
}
__device__ __forceinline__ bool div_den_ok(float den) { const float ad = __builtin_fabsf(den); return (ad >= 0x1p-20f) & (ad <= 0x1p20f); }
__device__ __forceinline__ bool div_num_ok(float x) { const float ax = __builtin_fabsf(x); return ((ax >= 0x1p-40f) & (ax <= 0x1p40f)) | (__float_as_uint(x) == 0u); }
__device__ __forceinline__ float div_by(const DivBy &q, float x)
{
    const float q0 = x * q.r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q.den, q0, x), q.r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-q.den, q1, x), q.r1, q1);
}

__device__ __forceinline__ f3 normalize(f3 a)                                                         // Vec3.h:100-103
{
    const float l2 = dot(a, a);
    // three quotients, one denominator: div_by when every active lane is inside its range -- every component at least 2^-40 in
    // magnitude and the squared length in [2^-36, 2^36], which puts the length in [2^-18, 2^18], the largest component in
    // [2^-19, 2^19], and the squared length inside the lean square root's range; a NaN component makes l2 NaN and fails the
    // compares.  Exact zeros, NaN and out-of-range vectors take the plain divides and the general square root, whole wave.
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const bool fast = (lo >= 0x1p-40f) & (l2 >= 0x1p-36f) & (l2 <= 0x1p36f);
    f3 r;
    if (__ballot(!fast) == 0ull) {
        const float l = (MIRT_TEXP & 2) ? __builtin_amdgcn_sqrtf(l2) : sqrt_in_range(l2);
        const DivBy by_l = div_prepare(l);
        r = mk(div_by(by_l, a.x), div_by(by_l, a.y), div_by(by_l, a.z));
    } else {
        const float l = MIRT_SQRT(l2);
        r = mk(a.x / l, a.y / l, a.z / l);
        if ((l == 0.0f) | __builtin_isinf(l2)) {              // both special cases behind one, almost never taken, branch
            const float q = (l == 0.0f) ? 0.0f : __builtin_nanf("");
            r = mk(q, q, q);
        }
    }
    return r;
}
__device__ __forceinline__ f3 ray_at(f3 o, f3 d, float t) { return o + scale(d, t); }                 // Ray.h:14

// ----------------------------------------------------------------------------- RNG source
// Philox4x32-10 (Salmon et al. SC'11) replaces std::random_device (Randomizer.h:11,22).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input xor in one instruction (v_bitop3_b32, truth table 0x96 = odd parity); the compiler leaves two v_xor
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The same block in two steps.  The counter is (c0, c1, c2, c3) = (pixel, sample, stage, iteration): round 1's products
// depend on (c0, c2) alone and c3 enters by one xor; of round 2's products one depends on round 1's first word, which the
// iteration has not reached yet, and so does one of round 3's.  The blocks of one (pixel, sample, stage) over many
// iterations therefore share up to 4 of their 20 multiplications: philox_head makes them once, philox_tail finishes a
// block for an iteration (rt_kernels.h coop_draws: the owner of a pending draw publishes its head).
// SHARE: 1 = round 1 shared (two products, four words), 2 = and round 2's (three, four words), 3 = and round 3's (four
// products, five words).
template <int SHARE> struct PhiloxHead { uint32_t w[SHARE >= 3 ? 5 : 4]; };
template <int SHARE>
__device__ __forceinline__ PhiloxHead<SHARE> philox_head(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1)
{
    static_assert(SHARE >= 1 && SHARE <= 3, "rounds that can be shared");
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;                    // round 1
    const uint32_t A = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96), D = (uint32_t)p0;
    PhiloxHead<SHARE> hd;
    hd.w[1] = (uint32_t)(p0 >> 32) ^ k1;                                // round 1's third word, before the iteration is xored in
    if constexpr (SHARE == 1) { hd.w[0] = (uint32_t)p1; hd.w[2] = A; hd.w[3] = D; }
    else {
        hd.w[0] = (uint32_t)p1 ^ (k0 + W0);                             // round 1's second word with round 2's key
        const uint64_t q0 = (uint64_t)0xD2511F53u * A;                  // round 2, the product that does not see the iteration
        const uint32_t E = __builtin_amdgcn_bitop3_b32((uint32_t)(q0 >> 32), D, k1 + W1, 0x96);
        if constexpr (SHARE == 2) { hd.w[2] = E; hd.w[3] = (uint32_t)q0; }
        else {
            const uint64_t q1 = (uint64_t)0xCD9E8D57u * E;              // round 3, likewise
            hd.w[2] = (uint32_t)q0 ^ (k1 + 2u * W1); hd.w[3] = (uint32_t)q1;
            hd.w[4] = (uint32_t)(q1 >> 32) ^ (k0 + 2u * W0);
        }
    }
    return hd;
}
template <int SHARE>
__device__ __forceinline__ void philox_tail(const PhiloxHead<SHARE> &hd, uint32_t it, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0, c1, c2, c3;
    if constexpr (SHARE == 1) { c0 = hd.w[2]; c1 = hd.w[0]; c2 = hd.w[1] ^ it; c3 = hd.w[3]; }
    else {
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (hd.w[1] ^ it);     // round 2's other product
        const uint32_t x0 = (uint32_t)(p1 >> 32) ^ hd.w[0], x1 = (uint32_t)p1;
        if constexpr (SHARE == 2) { c0 = x0; c1 = x1; c2 = hd.w[2]; c3 = hd.w[3]; }
        else {
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0;             // round 3's other product
            c0 = hd.w[4] ^ x1; c1 = hd.w[3]; c2 = (uint32_t)(p0 >> 32) ^ hd.w[2]; c3 = (uint32_t)p0;
        }
    }
    k0 += (uint32_t)SHARE * W0; k1 += (uint32_t)SHARE * W1;
#pragma unroll
    for (int r = SHARE; r < 10; ++r) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(m1 >> 32), c1, k0, 0x96);
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(m0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)m1; c3 = (uint32_t)m0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Randomizer.h:9-17 with (min,max) = (0,1): (float)((double)raw / 2^32).  u32 -> f32 conversion
// rounds to nearest-even exactly like double -> float of the exact quotient, and the power-of-two
// scale is exact, so no f64 is needed.  Can return 1.0f for raw >= 2^32-128, as the reference can.
__device__ __forceinline__ float map_unorm(uint32_t raw) { return (float)raw * 0x1p-32f; }

// Randomizer.h:20-26 with (min,max) = (-1,1): r = raw / 4294967295.0; r = r*2.0 + (-1.0); (float)r.
// Without the divide: q0 = raw * R with R = RN(1/4294967295.0) is the IEEE quotient or its neighbour, and that last bit of the
// double never reaches the float: (float)fma(q0, 2.0, -1.0) equals the reference's result for every one of the 2^32 words
// (tests/test_oracle.py walks them all against the true divide; R one ulp to either side fails for 20 and 44 of them).
// Rounds 1 and 2 refined q0 to the exact quotient first (Markstein, two more f64 fmas) -- exact, and not needed.  r*2.0 is
// exact, so fma(r,2,-1) rounds once, exactly as the separate multiply and add do.
__device__ __forceinline__ float map_pm1(uint32_t raw)
{
    const double R = 1.0 / 4294967295.0;
    return (float)__builtin_fma((double)raw * R, 2.0, -1.0);
}

#define MIRT_REJECT_CAP 64   // same cap as the oracle's ORC_REJECT_CAP

// Keyed draw source: counter = (pixel, sample, stage, block), key = seed.
// stage 0 = camera (block 0 = jitter u, jitter v, disk iteration 0; block k>=1 = disk iterations
// 2k-1, 2k); stage 1+depth = scatter at that depth (block i = rejection iteration i, lanes 0..2;
// the dielectric's single draw is block 0 lane 0).
struct PhiloxKeyed {
    // lazy: a stateless keyed source lets Lambertian skip its draws when the path is absorbed
    // anyway (Materials.cpp:32 decides independently of the draw); the result is unchanged.
    static constexpr bool lazy = true;
    uint32_t pixel, sample, stage, k0, k1;
    __device__ __forceinline__ void block(uint32_t b, uint32_t out[4]) const { philox4x32_10(pixel, sample, stage, b, k0, k1, out); }
};
// Explicit raw draws (function-level parity tests): consumed in call order.
struct StreamDraws {
    static constexpr bool lazy = false;   // consume draws exactly where the reference calls rd()
    const uint32_t *p; uint32_t n, pos;
    __device__ __forceinline__ uint32_t next() { uint32_t r = pos < n ? p[pos] : 0u; ++pos; return r; }
};

// Randomizer.h:40-48 (RomdomInUnitSphere)
__device__ __forceinline__ f3 in_unit_sphere(const PhiloxKeyed &g)
{
    f3 p = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t it = 0; it < MIRT_REJECT_CAP; ++it) {
        uint32_t r[4]; g.block(it, r);
        p = mk(map_pm1(r[0]), map_pm1(r[1]), map_pm1(r[2]));
        if (!(dot(p, p) >= 1.0f)) break;
    }
    return p;
}
__device__ __forceinline__ f3 in_unit_sphere(StreamDraws &g)
{
    f3 p = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t it = 0; it < MIRT_REJECT_CAP; ++it) {
        float x = map_pm1(g.next()), y = map_pm1(g.next()), z = map_pm1(g.next());
        p = mk(x, y, z);
        if (!(dot(p, p) >= 1.0f)) break;
    }
    return p;
}
__device__ __forceinline__ float scatter_unorm(const PhiloxKeyed &g) { uint32_t r[4]; g.block(0, r); return map_unorm(r[0]); }
__device__ __forceinline__ float scatter_unorm(StreamDraws &g) { return map_unorm(g.next()); }

// ----------------------------------------------------------------------------- scene records
// Device layout (packed from the SoA arrays of mirt_scene at upload):
struct PrimRec {            // 64 B; a sphere lives entirely in the first 32 B (one sector per divergent fetch)
    float    off[3];        // TranslatedInstance::m_offset
    uint32_t info;          // bit0 kind, bits1-2 aAxis, 3-4 bAxis, 5-6 cAxis, 7 reverse, 8 rotation active
    float    ga[3];         // sphere: r, 1.f/r, r*r ; rect: a0,a1,b0
    uint32_t mat;
    float    gb[2];         // rect: b1,c
    float    rot[6];        // sinx,cosx,siny,cosy,sinz,cosz
};
// A primitive in CHAIN FORM (include/mirt.h "general instance chains": a leaf with any nesting of TranslatedInstance /
// RotatedInstance around it) reuses the record: off = SphereHitable::m_center, info bit 9 set, bits 16-19 = number of ops,
// rot[0] = (as bits) index of its first OpRec; ops are stored outermost first.
struct OpRec { uint32_t kind; float a, b, c; };   // 16 B: MIRT_OP_TRANSLATE: offset a,b,c; MIRT_OP_ROTATE_X/Y/Z: sin a, cos b
struct Node2 {              // 64 B: both children's boxes in the parent, for near-first traversal
    float   lmn[3]; int32_t left;      // a node: its BYTE OFFSET in the array (index << 6); an object: ~index; MIRT_CHILD_NONE
    float   lmx[3]; int32_t right;
    float   rmn[3]; uint32_t lleaf;    // 1 when the left / right child is an object (or none): what slab_hit adds to t_max's bits,
    float   rmx[3]; uint32_t rleaf;    // stored so that no lane has to shift it out of the child index at every step
};
struct NodeRec {            // 32 B: one aligned 32-byte sector per (per-lane, divergent) node fetch
    float   mn[3]; int32_t left;
    float   mx[3]; int32_t right;
};
struct MatRec {             // 48 B
    uint32_t kind; float albedo[3];
    float    fuzz, ior, emit[3];
    int32_t  tex;           // image texture sampled for the albedo, -1 = the single colour above
    float    r0;            // Dielectric: ((1 - ior) / (1 + ior))^2 of Schlick (Optics.h:29-30), a per-material constant
    float    inv_ior;       // Dielectric: 1.0f / ior (Materials.cpp:193), likewise
};
struct TexRec { uint32_t w, h, offset, pad; };   // offset: first texel in the RGBA8 texel array
#define PI_KIND(i)   ((i) & 1u)
#define PI_AAX(i)    (((i) >> 1) & 3u)
#define PI_BAX(i)    (((i) >> 3) & 3u)
#define PI_CAX(i)    (((i) >> 5) & 3u)
#define PI_REV(i)    (((i) >> 7) & 1u)
#define PI_ROT(i)    (((i) >> 8) & 1u)
#define PI_CHAIN(i)  (((i) >> 9) & 1u)
#define PI_NOPS(i)   (((i) >> 16) & 15u)

struct Hit { float t; f3 pos, nrm; float u, v; int32_t mat, prim; };

// ----------------------------------------------------------------------------- AABB.cpp:5-21
// min/max are the Win32 compare-select macros (AABB.cpp:15-16), not fminf/fmaxf.
__device__ __forceinline__ bool aabb_hit(const float mn[3], const float mx[3], f3 o, f3 d, float t_min, float t_max)
{
#pragma unroll
    for (uint32_t a = 0; a < 3; ++a) {
        float invD = 1.0f / get(d, a);
        float t0 = (mn[a] - get(o, a)) * invD;
        float t1 = (mx[a] - get(o, a)) * invD;
        if (invD < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        t_min = (t0 > t_min) ? t0 : t_min;
        t_max = (t1 < t_max) ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// ----------------------------------------------------------------------------- Hitables.cpp
// RotatedInstance tables, Hitables.cpp:188-206
__device__ __forceinline__ void rot_axes(uint32_t ax, uint32_t &a, uint32_t &b, float &OP)
{
    if (ax == 0) { a = 1; b = 2; OP = -1.0f; }
    else if (ax == 1) { a = 0; b = 2; OP = 1.0f; }
    else { a = 0; b = 1; OP = -1.0f; }
}
// Hitables.cpp:244-250
__device__ __forceinline__ void rot_in(f3 &o, f3 &d, uint32_t ax, float s, float c)
{
    uint32_t a, b; float OP; rot_axes(ax, a, b, OP);
    float opp = -OP;
    f3 o0 = o, d0 = d;
    set(o, a, c * get(o0, a) + opp * s * get(o0, b));
    set(o, b, -opp * s * get(o0, a) + c * get(o0, b));
    set(d, a, c * get(d0, a) + opp * s * get(d0, b));
    set(d, b, -opp * s * get(d0, a) + c * get(d0, b));
}
// Hitables.cpp:253-260
__device__ __forceinline__ void rot_out(f3 &p, f3 &n, uint32_t ax, float s, float c)
{
    uint32_t a, b; float OP; rot_axes(ax, a, b, OP);
    f3 p0 = p, n0 = n;
    set(p, a, c * get(p0, a) + OP * s * get(p0, b));
    set(p, b, -OP * s * get(p0, a) + c * get(p0, b));
    set(n, a, c * get(n0, a) + OP * s * get(n0, b));
    set(n, b, -OP * s * get(n0, a) + c * get(n0, b));
}

// The ray in the primitive's leaf space: TranslatedInstance (Hitables.cpp:166) then, for a rect
// with a live rotation, RotZ, RotY, RotX (outermost first; SimpleObject.cpp:173-181).
// A rotation chain whose three angles are 0 is skipped: 1*x + (+-0)*y == x for finite input, so
// only the sign of a zero can differ.
// CHAIN (kernel variants for scenes that hold chain-form primitives; the others compile none of it): a primitive in chain form
// passes the ray through its ops, outermost first -- TranslatedInstance::Hit (Hitables.cpp:166) and RotatedInstance::Hit
// (:244-250) each transform the ray and call their child.
typedef float opf4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) opf4_t *OpPtr;            // one 16-byte global load per op
struct OpVal { uint32_t kind; float a, b, c; };
__device__ __forceinline__ OpVal load_op(const OpRec *ops, uint32_t i) { const opf4_t q = ((OpPtr)ops)[i]; OpVal v; v.kind = __float_as_uint(q.x); v.a = q.y; v.b = q.z; v.c = q.w; return v; }
template <bool CHAIN = false, class P>
__device__ __forceinline__ void to_leaf_space(const P &pr, uint32_t info, f3 &o, f3 &d, const OpRec *ops = nullptr)
{
    if (CHAIN && PI_CHAIN(info)) {
        const uint32_t first = __float_as_uint(pr.rot[0]), n = PI_NOPS(info);
        for (uint32_t k = 0; k < n; ++k) {
            const OpVal op = load_op(ops, first + k);
            if (op.kind == MIRT_OP_TRANSLATE) o = o - mk(op.a, op.b, op.c);
            else rot_in(o, d, op.kind - 1u, op.a, op.b);
        }
        return;
    }
    o = o - mk(pr.off[0], pr.off[1], pr.off[2]);
    if (PI_ROT(info)) {
        rot_in(o, d, 2, pr.rot[4], pr.rot[5]);
        rot_in(o, d, 1, pr.rot[2], pr.rot[3]);
        rot_in(o, d, 0, pr.rot[0], pr.rot[1]);
    }
}

// t of the closest accepted intersection of one primitive, or false.  Sphere: Hitables.cpp:20-43
// (centre 0, SimpleObject.cpp:136); rect: Hitables.cpp:95-100.  a_dd = dot(dir,dir) is passed in
// because it does not depend on the primitive when no rotation is live.
// by2a / div_ok (the near-first walks' sphere leaves): div_prepare(2 * dot(d, d)), made once per ray with r1 = NaN where 2a lies
// outside div_by's denominator range, and the wave-uniform promise t_min >= 2^-19.  The root's quotient is then div_by's five
// operations instead of the twelve of a full division, for the same bits: its first product |q0| <= 2^19 says that the
// numerator is below 2^40 (and nothing is NaN); a wave with a lane beyond that takes the plain division; a numerator below
// div_by's range gives |t| < 2^-19 < t_min by either route -- rejected, t unused.
template <bool CHAIN = false, class P>
__device__ __forceinline__ bool prim_t(const P &pr, f3 o, f3 d, float t_min, float t_max, float &t_out, const OpRec *ops = nullptr,
                                       const DivBy *by2a = nullptr, bool div_ok = false)
{
    uint32_t info = pr.info;
    to_leaf_space<CHAIN>(pr, info, o, d, ops);
    if (PI_KIND(info) == MIRT_PRIM_SPHERE) {
        f3 oc = o;                                                       // Hitables.cpp:20; the factory form's sphere sits at 0
        if (CHAIN && PI_CHAIN(info)) oc = o - mk(pr.off[0], pr.off[1], pr.off[2]);
        float a = dot(d, d);
        float b = 2.0f * dot(d, oc);
        float c = dot(oc, oc) - pr.ga[2];
        float disc = b * b - 4 * a * c;
        bool in = false;
        if (disc > 0) {                                                  // one exit, flags instead of early returns
            // The second root is tried when the first is out of [t_min, t_max] (Hitables.cpp:31-41), and it is never smaller than
            // the first: it can only be accepted when the first fell below t_min.  Which side of t_min the first root lies on is
            // decided by an estimate of its quotient wherever that is safe (within 3e-7 of it, the band around t_min is 1e-5 |t_min| wide),
            // so one IEEE divide runs instead of two; inside the band (or with a NaN) both run as written.  closest_list's
            // deferred roots do the same.
            const float sq = MIRT_SQRT(disc), two_a = 2.0f * a;
            const float n1 = -b - sq, n2 = -b + sq;
            const float q1 = n1 * (by2a ? by2a->r1 : __builtin_amdgcn_rcpf(two_a));   // (the refined reciprocal is as good an estimate)
            const float band = 0.00001f * __builtin_fabsf(t_min);                 // around t_min of either sign (the test hooks take any t_min)
            const bool first = q1 > t_min + band, second = q1 < t_min - band;
            const float num = first ? n1 : n2;
            float t;
            if (by2a && div_ok) {
                const float q0 = num * by2a->r1;
                const unsigned long long beyond = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(q0) <= 0x1p19f));
                uint32_t slow = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)beyond | (uint32_t)(beyond >> 32)));
                asm volatile("" : "+s"(slow));                          // a branch on a scalar, not both forms and a select
                if (__builtin_expect(slow == 0u, 1)) {
                    const float qr = __builtin_fmaf(__builtin_fmaf(-two_a, q0, num), by2a->r1, q0);      // div_by, from its first product on
                    t = __builtin_fmaf(__builtin_fmaf(-two_a, qr, num), by2a->r1, qr);
                } else {
                    asm volatile("; plain division for the whole wave");
                    t = num / two_a;
                }
            } else t = num / two_a;
            in = (t <= t_max) & (t >= t_min);
            if (!(first | second)) {
                t = n1 / two_a;
                in = (t <= t_max) & (t >= t_min);
                if (!in) { t = n2 / two_a; in = (t <= t_max) & (t >= t_min); }
            }
            t_out = t;
        }
        return in;
    }
    uint32_t aA = PI_AAX(info), bA = PI_BAX(info), cA = PI_CAX(info);
    float t = (pr.gb[1] - get(o, cA)) / get(d, cA);
    float a = get(o, aA) + t * get(d, aA);
    float b = get(o, bA) + t * get(d, bA);
    t_out = t;
    return (t <= t_max) & (t >= t_min) & (a >= pr.ga[0]) & (a <= pr.ga[1]) & (b >= pr.ga[2]) & (b <= pr.gb[0]);
}

// ----------------------------------------------------------------------------- atan2f / asinf of CalculateUV, pinned by definition
// The reference calls the MSVC CRT's atan2f and asinf (Hitables.cpp:70-71), whose bits are not known here; glibc's (the
// oracle's host) and the device library's differ from each other in the last place, which used to show up as a texel's
// width of difference on textured spheres.  Like Schlick's pow, both sides therefore evaluate ONE stated sequence of IEEE
// float operations (single-precision minimax polynomials after the classic argument reductions, error below 3 ulp --
// closer to any correctly working libm than two libms are to each other): the oracle and the device agree bit for bit.
__device__ __forceinline__ float det_atanf_pos(float x)                       // x >= 0 (or NaN)
{
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }            // tan(3 pi / 8): pi/2 - atan(1/x)
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }   // tan(pi / 8): pi/4 + atan((x-1)/(x+1))
    const float z = x * x;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return y + p;
}
__device__ __forceinline__ float det_atan2f(float y, float x)
{
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966f;
    if (x != x || y != y) return __builtin_nanf("");
    if (x == 0.0f) return y > 0.0f ? PIO2 : (y < 0.0f ? -PIO2 : 0.0f);
    if (y == 0.0f) return x < 0.0f ? PI : 0.0f;
    const float q = y / x;
    const float a = det_atanf_pos(__builtin_fabsf(q));
    const float z = q < 0.0f ? -a : a;                                            // atan is odd
    const float w = x < 0.0f ? (y < 0.0f ? -PI : PI) : 0.0f;
    return w + z;
}
__device__ __forceinline__ float det_asinf(float x)
{
    const float PIO2 = 1.5707963267948966f;
    const float a = __builtin_fabsf(x);
    if (!(a <= 1.0f)) return __builtin_nanf("");                                  // outside the domain (and NaN), as asinf answers
    if (a < 1.0e-4f) return x;
    float z, s;
    const bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); s = __builtin_sqrtf(z); } else { s = a; z = s * s; }
    float r = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    if (big) { r = r + r; r = PIO2 - r; }
    return x < 0.0f ? -r : r;
}

// The HitRecord of an accepted intersection (Hitables.cpp:35-39,102-109, then the instance chain
// on the way out: :253-260 per rotation, :169 translate).  Recomputed from (prim, t): every step
// is a deterministic function of the same inputs, so the bits equal the reference's in-place
// record.  with_uv additionally evaluates CalculateUV (Hitables.cpp:62-78) / the rect's u,v.
template <bool CHAIN = false, class P>
__device__ __forceinline__ void prim_record(const P &pr, int32_t prim, f3 o, f3 d, float t, bool with_uv, Hit &h, const OpRec *ops = nullptr)
{
    uint32_t info = pr.info;
    const bool chain = CHAIN && PI_CHAIN(info);
    to_leaf_space<CHAIN>(pr, info, o, d, ops);
    h.t = t; h.mat = (int32_t)pr.mat; h.prim = prim; h.u = 0.0f; h.v = 0.0f;
    h.pos = ray_at(o, d, t);
    if (PI_KIND(info) == MIRT_PRIM_SPHERE) {
        f3 rel = h.pos;                                         // pos - m_center, Hitables.cpp:37,65
        if (chain) rel = h.pos - mk(pr.off[0], pr.off[1], pr.off[2]);
        h.nrm = scale(rel, pr.ga[1]);                           // (pos - centre) / r == * (1.f/r), Vec3.h:85-88
        if (with_uv) {
            f3 p = normalize(rel);
            float theta = det_atan2f(p.z, p.x);
            float phi = det_asinf(p.y);
            const float PI = 3.14159265358979323846f;
            h.u = 1.0f - (theta + PI) / (2.0f * PI);
            h.v = (phi + PI / 2.0f) / PI;
        }
    } else {
        uint32_t aA = PI_AAX(info), bA = PI_BAX(info), cA = PI_CAX(info);
        if (with_uv) {
            float a = get(o, aA) + t * get(d, aA);
            float b = get(o, bA) + t * get(d, bA);
            h.u = (a - pr.ga[0]) / (pr.ga[1] - pr.ga[0]);
            h.v = (b - pr.ga[2]) / (pr.gb[0] - pr.ga[2]);
        }
        const float sgn = PI_REV(info) ? -1.0f : 1.0f;                   // selects, not the branches of set()
        h.nrm = mk(cA == 0 ? sgn : 0.0f, cA == 1 ? sgn : 0.0f, cA >= 2 ? sgn : 0.0f);
        if (!chain && PI_ROT(info)) {
            rot_out(h.pos, h.nrm, 0, pr.rot[0], pr.rot[1]);
            rot_out(h.pos, h.nrm, 1, pr.rot[2], pr.rot[3]);
            rot_out(h.pos, h.nrm, 2, pr.rot[4], pr.rot[5]);
        }
    }
    if (chain) {                                                // the record on its way back out, innermost instance first (:169, :253-260)
        const uint32_t first = __float_as_uint(pr.rot[0]), n = PI_NOPS(info);
        for (uint32_t k = n; k-- > 0;) {
            const OpVal op = load_op(ops, first + k);
            if (op.kind == MIRT_OP_TRANSLATE) h.pos = h.pos + mk(op.a, op.b, op.c);
            else rot_out(h.pos, h.nrm, op.kind - 1u, op.a, op.b);
        }
        return;
    }
    h.pos = h.pos + mk(pr.off[0], pr.off[1], pr.off[2]);
}

// ----------------------------------------------------------------------------- Optics.h
__device__ __forceinline__ f3 reflect(f3 v, f3 n) { return v - scale(n, 2 * dot(v, n)); }            // Optics.h:8-12
// dt: dot(v, n), which the caller may already hold (see scatter_finish)
__device__ __forceinline__ bool refract_dt(f3 v, f3 n, float dt, float ni_over_nt, f3 &refracted)     // Optics.h:16-22
{
    float disc = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    bool ret = disc > 0;
    if (ret) refracted = scale(v - scale(n, dt), ni_over_nt) - scale(n, MIRT_SQRT(disc));
    return ret;
}
__device__ __forceinline__ bool refract(f3 v, f3 n, float ni_over_nt, f3 &refracted)                 // Optics.h:14-23
{
    float dt = dot(v, n);
    float disc = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    bool ret = disc > 0;
    if (ret) refracted = scale(v - scale(n, dt), ni_over_nt) - scale(n, MIRT_SQRT(disc));
    return ret;
}
// Optics.h:27-32; pow(float,int) -> x^5 in double rounded once, as the oracle defines it.
__device__ __forceinline__ float pow5f(float x) { double d = x, d2 = d * d, d4 = d2 * d2; return (float)(d4 * d); }
__host__ __device__ __forceinline__ float schlick_r0(float ref_idx)
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    return r0 * r0;
}
__device__ __forceinline__ float schlick(float cosine, float r0) { return r0 + (1 - r0) * pow5f(1 - cosine); }

// ----------------------------------------------------------------------------- Materials.cpp
// Scatter is split in three so the render kernel can draw for a whole wavefront at once:
//   scatter_need   what the material wants from the RNG (nothing / a unit-sphere point / one unorm)
//   (the draws)    per lane in the function-level hooks, wave-cooperative in k_bounce
//   scatter_finish the rest of Scatter, given the draws
struct Scat { bool alive, emits; f3 atten, dir, emitted; };
enum { NEED_NONE = 0, NEED_SPHERE = 1, NEED_UNORM = 2 };

// lazy: with a stateless keyed source an absorbed Lambertian path (Materials.cpp:32 decides
// independently of the draw) skips its draws; the explicit-stream hooks never skip.
__device__ __forceinline__ uint32_t scatter_need(uint32_t kind, f3 dir_in, f3 nrm, bool lazy)
{
    if ((MIRT_TEXP & 1) && kind != MIRT_MAT_DIFFUSE_LIGHT) kind = MIRT_MAT_LAMBERTIAN;
    const bool front = dot(dir_in, nrm) < 0;                                 // selects: the kinds of a wave are mixed anyway
    const bool sphere = (kind == MIRT_MAT_METAL) | ((kind == MIRT_MAT_LAMBERTIAN) & (front | !lazy));
    return sphere ? NEED_SPHERE : (kind == MIRT_MAT_DIELECTRIC ? NEED_UNORM : NEED_NONE);
}

// SimpleTexture2D_TGAImage::Sample, SimpleTexture2D.cpp:130-147: nearest texel; float -> UINT32 as MSVC x64 does it
// (cvttss2si to 64 bits, low half kept: (-1,0) -> 0, NaN / out of range -> 0), `> size-1` clamps, channel / 255.0f.
__device__ __forceinline__ uint32_t f2u32_msvc(float f)
{
    if (!(f > -9223372036854775808.0f && f < 9223372036854775808.0f)) return 0u;
    return (uint32_t)(unsigned long long)(long long)f;
}
__device__ __forceinline__ f3 tex_sample(const TexRec &t, const uint32_t *texels, float u, float v)
{
    uint32_t i = f2u32_msvc(u * (float)t.w);
    uint32_t j = f2u32_msvc((1.0f - v) * (float)t.h - 0.001f);
    if (i > t.w - 1u) i = t.w - 1u;
    if (j > t.h - 1u) j = t.h - 1u;
    const uint32_t px = texels[(size_t)t.offset + (size_t)j * t.w + i];
    return mk((float)(px & 0xFFu) / 255.0f, (float)((px >> 8) & 0xFFu) / 255.0f, (float)((px >> 16) & 0xFFu) / 255.0f);
}
// m_albedo->Sample(rec.m_u, rec.m_v), Materials.cpp:30,104
template <class M>
__device__ __forceinline__ f3 albedo_of(const M &m, const TexRec *texs, const uint32_t *texels, float u, float v)
{
    if (m.tex >= 0) return tex_sample(texs[m.tex], texels, u, v);
    return mk(m.albedo[0], m.albedo[1], m.albedo[2]);
}

// albedo: what the material's texture returns at the hit (albedo_of, or the single colour when the scene has no images)
// DEFAULTS: also give the fields a material does not touch their values of the reference (zero: the caller's
// `emmitted.zero()`, HomemadeRayTracer.cpp:173, and the default-constructed outputs).  The render kernel passes false
// and reads atten/dir only when alive, emitted only when emits -- which saves it the register moves.
// PART: 0 the whole of Scatter; 1 without the metal/dielectric branch, 2 only that branch (the render kernel's second,
// material-sorted stream shades those two kinds apart from the rest; the caller routes each vertex to the right part)
// lambert_front: >= 0 when the caller already knows Lambertian's `dot(r_in.direction, rec.normal) < 0` (Materials.cpp:32) -- the
// render kernel does: with the lazy keyed draws scatter_need asks for a sphere point exactly when that holds -- so the dot
// product is not taken a second time (five instructions per pass); -1: evaluate it here.
template <bool DEFAULTS = true, int PART = 0, class M>
__device__ __forceinline__ void scatter_finish(const M &m, f3 albedo, f3 dir_in, const f3 &pos, const f3 &nrm, f3 sphere_pt, float unorm, Scat &o, int lambert_front = -1)
{
    if (DEFAULTS) {
        o.emitted = mk(0.0f, 0.0f, 0.0f);                               // HomemadeRayTracer.cpp:173
        o.atten = mk(0.0f, 0.0f, 0.0f);
        o.dir = mk(0.0f, 0.0f, 0.0f);
    }
    o.emits = false;
    uint32_t kind = m.kind;
    if ((MIRT_TEXP & 1) && kind != MIRT_MAT_DIFFUSE_LIGHT) kind = MIRT_MAT_LAMBERTIAN;
    if (PART != 2 && kind == MIRT_MAT_LAMBERTIAN) {                     // Materials.cpp:19-33
        o.alive = lambert_front >= 0 ? lambert_front != 0 : dot(dir_in, nrm) < 0;
        o.atten = albedo;
        f3 target = (pos + nrm) + sphere_pt;
        o.dir = target - pos;
    } else if (PART != 1 && (PART == 2 || kind == MIRT_MAT_METAL || kind == MIRT_MAT_DIELECTRIC)) {
        f3 uv = normalize(dir_in);                                      // shared by both (Materials.cpp:102,175)
        f3 refl = reflect(uv, nrm);                                     // Materials.cpp:102,209
        if (kind == MIRT_MAT_METAL) {                                   // Materials.cpp:99-106
            o.dir = refl + scale(sphere_pt, m.fuzz);
            o.atten = albedo;
            o.alive = dot(o.dir, nrm) > 0;
        } else {                                                        // Materials.cpp:172-217
            float ior = m.ior;
            f3 outward, refracted = mk(0.0f, 0.0f, 0.0f);
            float ni_over_nt, cosine, reflect_prob;
            o.atten = mk(1.0f, 1.0f, 1.0f);
            float dn = dot(uv, nrm);
            if (dn > 0) { outward = neg(nrm); ni_over_nt = ior; cosine = ior * dn; }
            else { outward = nrm; ni_over_nt = m.inv_ior; cosine = -dn; }
            // dot(uv, outward) without a second dot product: outward is nrm or 0 - nrm, and with dn > 0 the products of the
            // negated normal are the negated products (a zero component only changes the sign of a zero term, which a non-zero
            // sum does not see), so the sum is -dn bit for bit; with dn <= 0 (or NaN) it is the same expression as dn
            const float dt = dn > 0 ? -dn : dn;
            if (refract_dt(uv, outward, dt, ni_over_nt, refracted)) reflect_prob = schlick(cosine, m.r0);
            else reflect_prob = 1.0f;
            o.dir = (unorm < reflect_prob) ? refl : refracted;
            o.alive = true;
        }
    } else {                                                            // DiffuseLight, Materials.cpp:260-264
        o.emitted = mk(m.emit[0], m.emit[1], m.emit[2]);
        o.alive = false; o.emits = true;
    }
}

// one lane, draws in the reference's call order
template <class M, class G>
__device__ __forceinline__ void scatter(const M &m, f3 albedo, f3 dir_in, const f3 &pos, const f3 &nrm, G &g, Scat &o)
{
    uint32_t need = scatter_need(m.kind, dir_in, nrm, G::lazy);
    f3 sp = mk(0.0f, 0.0f, 0.0f); float u = 0.0f;
    if (need == NEED_SPHERE) sp = in_unit_sphere(g);
    else if (need == NEED_UNORM) u = scatter_unorm(g);
    scatter_finish(m, albedo, dir_in, pos, nrm, sp, u, o);
}

// ----------------------------------------------------------------------------- SimpleCamera.cpp:55-60
__device__ __forceinline__ void camera_ray(const mirt_camera &c, float u, float v, f3 disk, f3 &org, f3 &dir)
{
    f3 rd = scale(disk, c.lens_radius);
    f3 cu = mk(c.u[0], c.u[1], c.u[2]), cv = mk(c.v[0], c.v[1], c.v[2]);
    f3 offset = scale(cu, rd.x) + scale(cv, rd.y);
    f3 origin = mk(c.origin[0], c.origin[1], c.origin[2]);
    f3 d = mk(c.top_left[0], c.top_left[1], c.top_left[2]) + scale(mk(c.horizontal[0], c.horizontal[1], c.horizontal[2]), u);
    d = d + scale(mk(c.vertical[0], c.vertical[1], c.vertical[2]), v);
    d = d - origin;
    d = d - offset;
    org = origin + offset;
    dir = d;
}

} // namespace mirt
