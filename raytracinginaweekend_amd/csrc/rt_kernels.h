// rt_kernels.h -- the gfx950 kernels of the path tracer.
//
// One batch = a run of pixels x a run of samples, one path per (pixel, sample).  Two engines trace it, with the
// same per-path arithmetic and Philox keys (identical images):
//
//   k_trace / k_trace_bvh   (default) persistent, one path per lane from camera ray to termination; a finished
//                           lane takes a fresh camera ray from its wave's LDS stash.  No path state in HBM.
//   k_bounce<FIRST>, k_bounce   the wavefront formulation: one launch per depth over compacted path queues
//                           (three float4 planes per slot, ballot + prefix popcount, chunked slot reservations).
//
//   k_accumulate      ordered per-pixel sum of the batch's per-sample radiance
//   k_resolve         mean, sqrt gamma, clamp, RGBA8  (once per image)
//
// A path contributes radiance exactly once, at the vertex where it ends (emission is non-zero
// only on DiffuseLight, which never scatters, Materials.cpp:260-264, and on a miss,
// HomemadeRayTracer.cpp:186), so each path writes one 12-byte slot of rad[] when it terminates and
// k_accumulate adds the samples of a pixel in sample order -- the same order as the reference's
// `col += Sample(r, 0)` loop (HomemadeRayTracer.cpp:126-133).  No float atomics, no run-to-run
// variation, and no dependence on which lane or launch traced a path.
#pragma once
#include "rt_device.h"

namespace mirt {

#ifndef MIRT_BLOCK
#define MIRT_BLOCK 256
#endif
#ifndef MIRT_LEAF_DIVBY
#define MIRT_LEAF_DIVBY 1   // sphere leaves of the near-first walks: the root's quotient through div_by (rt_device.h prim_t)
#endif
#ifndef MIRT_LEAF_WANT
#define MIRT_LEAF_WANT 4u   // closest_bvh2: lanes waiting on a leaf that start an object-test pass (option leaf_wait)
#endif
#ifndef MIRT_BVH_WAVES
#define MIRT_BVH_WAVES 8   // waves per SIMD the BVH kernels are compiled for (register cap 512 / MIRT_BVH_WAVES)
#endif

// A path queue: three planes of float4 per slot -- (ox,oy,oz,dx) (dy,dz,tr,tg) (tb,pixel,sample,-) -- so a
// path moves with three 16-byte accesses per lane, each perfectly coalesced (lane i at base + 16*i).
struct PathQ {
    float4 *p0, *p1, *p2;
};

// A path's radiance slot: three floats, 12 bytes.  Written once, by the lane that ends the path; read once, by k_accumulate,
// which is bound by HBM -- so the slot carries no padding word.
struct RadSlot { float r, g, b; };
typedef float f32x3_t __attribute__((ext_vector_type(3)));
typedef f32x3_t f32x3_a4 __attribute__((aligned(4)));
__device__ __forceinline__ void store_rad(RadSlot *rad, size_t slot, f3 v)
{
    f32x3_t x; x.x = v.x; x.y = v.y; x.z = v.z;
    *(f32x3_a4 *)(rad + slot) = x;
}

struct SceneDev {
    const PrimRec *prims;  uint32_t n_prims;
    const PrimRec *lprims; uint32_t lend[4];     // the list walk's copy, sorted by class (closest_list); .mat = index in `prims`
    uint32_t list_fast;                           // bit 0: every unrotated rect has |plane|, |offset| <= 2^38: rect_loop may share its divisions;
                                                  // bits 1-3: orientation class 0-2 starts with a facing pair (rect_pair)
    const NodeRec *nodes;  uint32_t n_nodes;     // the tree as given (or built), walked in the reference's order
    const Node2   *nodes2;                        // same tree, child boxes in the parent, walked near-first
    uint32_t node_bytes;                          // size of the node record the launch stages in LDS (32 or 64)
    uint32_t single;                              // every object is one primitive: object index == primitive index
    const float4  *sph4;                          // scenes of single unrotated spheres only (else NULL): (centre, r*r) per object -- all a leaf test reads, in ONE 16-byte access
    const uint2   *objs;   uint32_t n_objects;   // (first, count)
    const MatRec  *mats;   uint32_t n_mats;
    const OpRec   *ops;                           // instances of chain-form primitives (rt_device.h OpRec), NULL when the scene has none
    int32_t root;
    float   ambient[3];
    // dynamic-LDS plan (in records): [prims][nodes][mats][stack]
    uint32_t lds_prims, lds_nodes, lds_mats, stack_depth;
    // traversal stack: levels [0, stack_lds) live in LDS ([level][lane]); deeper levels -- rare with a good tree
    // walked near-first -- go to `spill` in HBM, [level - stack_lds][thread of the grid].  Keeping the LDS part
    // short is what lets 8 workgroups share a CU.
    uint32_t stack_lds, spill_stride;
    int32_t *spill;
    // image textures (SimpleTexture2D_TGAImage): per-texture size/offset and all texels as packed RGBA8 words
    const TexRec *texs; const uint32_t *texels; uint32_t n_textures;
};

struct BounceArgs {
    PathQ in, out;
    // Queues are reserved in chunks of 1 << chunk_shift slots: a workgroup takes a chunk from the
    // out-queue with ONE atomic and packs the survivors of its successive tiles into it, so the
    // reservation counter sees one atomic per ~chunk/256 tiles instead of one per tile (a single
    // word serves only ~90 M returning atomics/s, which capped the first version of this kernel).
    // len[c] = slots of chunk c that hold paths; the unused tail of a chunk is skipped by the reader.
    const uint32_t *res_in;        // slots reserved in the in-queue (multiple of the chunk size)
    const uint32_t *len_in;
    uint32_t *cursor;              // next unclaimed work unit of the in-queue: workgroups claim units (a chunk, or a
                                   // power-of-two part of one when the queue is short) dynamically, so a launch
                                   // balances itself whatever the grid size and residency turn out to be
    uint32_t *res_out;
    uint32_t *len_out;
    uint32_t *alive_out;           // paths written to the out-queue (statistics: rays of the next depth)
    uint32_t *overflow;            // set if a reservation would pass cap_slots (never, by the host's bound)
    uint32_t chunk_shift, cap_slots;
    uint32_t n_first;
    RadSlot *rad;
    SceneDev sc;
    mirt_camera cam;
    uint32_t width, height, pix0, npix, spp0;
    uint32_t depth, max_depth, flags, k0, k1;
    FastDiv by_width;              // pixel / width (rt_device.h fast_div)
};

// MIRT_EXP: ablation bits for timing experiments on the depth>=1 kernel (tools/ablate.sh); never set in the product
#ifndef MIRT_EXP
#define MIRT_EXP 0
#endif
#define EXP_ON(bit) (!FIRST && (MIRT_EXP & (bit)))

// ---------------------------------------------------------------------------------- LDS staging
// Stride of a primitive record in the workgroup's LDS copy.  The record is 64 bytes = 16 banks, so with a 64-byte stride the
// records of primitives p, p+4, p+8 start on the same bank: lanes of one wave that fetch the records of their (different)
// closest primitives collided -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 16 % in k_trace on the Cornell box (9 primitives:
// three triples/pairs share banks).  With one 16-byte chunk of padding per record (20 banks) the first 16 records start on 16
// different bank groups.  -DMIRT_LDS_PRIM_STRIDE=64 restores the old layout for the A/B (profiles/r03_lds_conflicts.txt).
#ifndef MIRT_RECT_PAIRS
#define MIRT_RECT_PAIRS 1         // facing pairs of rects tested as one (rect_pair)
#endif
#ifndef MIRT_LDS_PRIM_STRIDE
#define MIRT_LDS_PRIM_STRIDE 80u
#endif
struct LdsView {
    PrimRec *prims; unsigned char *nodes; MatRec *mats; int32_t *stack;
};
// A wave-uniform kernel argument made a plain SGPR value: without this the compiler re-reads arguments of a hot loop from
// the kernarg segment in every pass (s_load + s_waitcnt in front of the node fetch) rather than keep them in registers.
template <class T> __device__ __forceinline__ T sgpr_pin(T x) { asm volatile("" : "+s"(x)); return x; }
// A vector register's present contents as a defined value (no instruction): for variables that only some lanes ever set and read.
__device__ __forceinline__ uint32_t any_vgpr() { uint32_t x; asm("" : "=v"(x)); return x; }
struct Stack {                       // one lane's traversal stack (see SceneDev::stack_lds)
    int32_t *lds, *hbm; uint32_t n_lds, stride, sp;   // capacity: tree depth + 1 levels (SceneDev::stack_depth), which no walk exceeds
    __device__ __forceinline__ Stack(const SceneDev &sc, const LdsView &v)
        : lds(v.stack + threadIdx.x), hbm(sc.spill + (blockIdx.x * MIRT_BLOCK + threadIdx.x)), n_lds(sgpr_pin(sc.stack_lds)),
          stride(sgpr_pin(sc.spill_stride)), sp(0) {}
    __device__ __forceinline__ void push(int32_t x)
    {
        if (sp < n_lds) lds[sp * MIRT_BLOCK] = x; else hbm[(size_t)(sp - n_lds) * stride] = x;
        ++sp;
    }
    __device__ __forceinline__ int32_t pop()
    {
        // two address spaces spelled out: the plain conditional compiles to ONE flat load of a selected address, and a flat load
        // goes through the texture addresser even when it ends in LDS
        --sp;
        int32_t x;
        if (sp < n_lds) x = ((const __attribute__((address_space(3))) int32_t *)lds)[sp * MIRT_BLOCK];
        else x = ((const __attribute__((address_space(1))) int32_t *)hbm)[(size_t)(sp - n_lds) * stride];
        return x;
    }
};
__device__ __forceinline__ LdsView lds_view(const SceneDev &sc, unsigned char *base)
{
    LdsView v;
    v.prims = (PrimRec *)base;
    v.nodes = base + MIRT_LDS_PRIM_STRIDE * sc.lds_prims;
    v.mats = (MatRec *)(v.nodes + sc.node_bytes * sc.lds_nodes);
    v.stack = (int32_t *)((unsigned char *)v.mats + sizeof(MatRec) * sc.lds_mats);
    return v;
}
__device__ __forceinline__ void lds_copy16(void *dst, const void *src, uint32_t bytes)
{
    const uint4 *s = (const uint4 *)src; uint4 *d = (uint4 *)dst;
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ void lds_stage(const SceneDev &sc, const LdsView &v)
{
    {   // primitive records, MIRT_LDS_PRIM_STRIDE bytes apart (four 16-byte chunks of payload each)
        const uint4 *src = (const uint4 *)sc.prims; uint4 *dst = (uint4 *)v.prims;
        for (uint32_t i = threadIdx.x; i < 4u * sc.lds_prims; i += blockDim.x) dst[(i >> 2) * (MIRT_LDS_PRIM_STRIDE / 16u) + (i & 3u)] = src[i];
    }
    lds_copy16(v.nodes, sc.node_bytes == sizeof(Node2) ? (const void *)sc.nodes2 : (const void *)sc.nodes, sc.node_bytes * sc.lds_nodes);
    lds_copy16(v.mats, sc.mats, sizeof(MatRec) * sc.lds_mats);
    __syncthreads();
}

// The scene's primitive records as a kernel variant reads them: the workgroup's LDS copy (PLDS, padded stride) or memory.
template <bool PLDS>
struct Prims {
    const unsigned char *base;
    __device__ __forceinline__ Prims(const SceneDev &sc, const LdsView &v) : base(PLDS ? (const unsigned char *)v.prims : (const unsigned char *)sc.prims) {}
    __device__ __forceinline__ const PrimRec &operator[](uint32_t p) const
    {
        if constexpr (PLDS) return *(const PrimRec *)(base + __umul24(p, MIRT_LDS_PRIM_STRIDE));
        else return ((const PrimRec *)base)[p];
    }
};

// ---------------------------------------------------------------------------------- closest hit
// List: Hitables.cpp:130-142 semantics over every object in order (= every primitive in order):
// t_max shrinks on each accepted hit, `<=` lets a later primitive win a tie.  All lanes walk the
// same primitive, so a record is wave-uniform: it comes through the scalar cache into SGPRs.
// Three things are done differently from a literal walk, none changes a bit of the result:
//  * the walk goes over a copy sorted by class, one loop per class (see closest_list below), with the reference's
//    sequential rule "t <= t_max, later index wins a tie" restated with the scene index so that order does not
//    matter: t < t_max || (t == t_max && index > best);
//  * rect tests have their axis triple fixed at compile time and no early-outs;
//  * a sphere's roots (IEEE sqrt + IEEE divide, ~45 instructions that only the few lanes whose line meets the
//    sphere need) are deferred: the lane parks (b, disc, index) and the roots are evaluated once after the
//    loop -- or earlier, when a lane that already has a parked sphere meets another one.  Which root is taken
//    does not depend on t_max (the second root is never smaller than the first).
typedef uint32_t ListWord4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) ListWord4 *ListPtr;   // records of the sorted list, read through the scalar cache
// A facing pair: the lowest and the highest plane of an orientation class (Cornell's left and right wall, its floor and ceiling).
// A ray whose origin lies between the two planes can only meet the one ahead of it: for the other, t = (k - origin.c) / direction.c
// is negative or zero and fails t >= t_min.  When that holds for every lane of the wave, ONE test serves both rects: each lane
// takes the record of the rect ahead from the workgroup's LDS copy (a per-lane index, where the class loop reads wave-uniform
// records through the scalar cache) and runs the very arithmetic of rect_loop on it; the numerators, which decide "behind",
// are computed as rect_loop computes them, so the argument holds in float arithmetic and not just on paper.  Otherwise
// (some lane outside the slab and heading in, or a NaN) nothing happens here and the class loop tests both as usual.  FAST path
// only: direction.c is then non-zero and t_min >= 2^-19 > 0.
// The pair's records come first in their class's run of the sorted list, low plane first; the upload leaves what this function
// needs of both in their unused rotation words: (k, offset.c) of the low and of the high one in the first record, the two scene
// indices in the second.
template <int A, int B, int C, bool PLDS>
__device__ __forceinline__ void rect_pair(const Prims<PLDS> &P, const DivBy &by_dc, ListPtr &q, uint32_t &p, f3 o, f3 d,
                                          float t_min, float &t_max, int32_t &best)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x4 pf = q[3];
    const u32x2 pi = *(const __attribute__((address_space(4))) u32x2 *)(q + 7);
    const float oc = C == 0 ? o.x : (C == 1 ? o.y : o.z), dc = C == 0 ? d.x : (C == 1 ? d.y : d.z);
    const float num_lo = __uint_as_float(pf.x) - (oc - __uint_as_float(pf.y)), num_hi = __uint_as_float(pf.z) - (oc - __uint_as_float(pf.w));
    // the rect behind the ray must fail t >= t_min for sure: its numerator has the sign that makes t <= 0, or -- an origin ON a
    // wall lands an ulp to either side of it -- is at most half of t_min * |direction.c| (division is monotone, so t <= ~t_min / 2)
    const bool fwd = dc > 0.0f;
    const bool behind_misses = (fwd ? num_lo : -num_hi) <= (0.5f * t_min) * __builtin_fabsf(dc);
    if (__ballot(!behind_misses) != 0ull) return;
    const int32_t idx = fwd ? (int32_t)pi.y : (int32_t)pi.x;
    const float num = fwd ? num_hi : num_lo;
    const PrimRec &r = P[(uint32_t)idx];
    const float da = A == 0 ? d.x : (A == 1 ? d.y : d.z), db = B == 0 ? d.x : (B == 1 ? d.y : d.z);
    const float oa = (A == 0 ? o.x : (A == 1 ? o.y : o.z)) - r.off[A], ob = (B == 0 ? o.x : (B == 1 ? o.y : o.z)) - r.off[B];
    const float t = div_by(by_dc, num);
    const float a = oa + t * da;
    const float b = ob + t * db;
    const bool in_t = (t >= t_min) & ((t < t_max) | ((t == t_max) & (idx > best)));
    const bool in_ab = (a >= r.ga[0]) & (a <= r.ga[1]) & (b >= r.ga[2]) & (b <= r.gb[0]);
    const bool hit = in_t & in_ab;
    t_max = hit ? t : t_max;
    best = hit ? idx : best;
    p += 2; q += 8;
}

// One class of the sorted list: unrotated rects whose axis triple is (A, B, C).  Branch-free on purpose: with 64
// unrelated rays some lane always passes the t test, so skipping a/b buys nothing and the selects keep the loop
// free of phi copies.  pr.mat of the sorted copy holds the primitive's index in the scene's own order.
// FAST: t = (k - origin.c) / direction.c through div_by (rt_device.h), five operations instead of the twelve of a full
// IEEE divide because every rect of the class divides by the same direction.c.  closest_list takes it when the whole wave
// is inside div_by's operand range (see there); a numerator too small for its promise gives |t| < 2^-19 <= t_min
// either way, so the rect is missed by both forms.
template <int A, int B, int C, bool FAST, bool PLDS = false>
__device__ __forceinline__ void rect_loop(ListPtr &q, uint32_t &p, uint32_t end, f3 o, f3 d,
                                          float t_min, float &t_max, int32_t &best, uint32_t list_fast = 0u, const Prims<PLDS> *P = nullptr)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const float dc = C == 0 ? d.x : (C == 1 ? d.y : d.z);
    const float da = A == 0 ? d.x : (A == 1 ? d.y : d.z), db = B == 0 ? d.x : (B == 1 ? d.y : d.z);
    if (p >= end) return;
    const DivBy by_dc = div_prepare(dc);
    if constexpr (FAST && PLDS && MIRT_RECT_PAIRS) {
        constexpr int CLS = C == 2 ? 0 : (C == 1 ? 1 : 2);
        if (list_fast & (2u << CLS)) rect_pair<A, B, C, PLDS>(*P, by_dc, q, p, o, d, t_min, t_max, best);
    }
    for (; p < end; ++p, q += 4) {                                          // one running record pointer through all the class loops
        const u32x4 q0 = q[0], q1 = q[1];
        const u32x2 q2 = *(const __attribute__((address_space(4))) u32x2 *)(q + 2);
        const f3 ol = o - mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const int32_t idx = (int32_t)q1.w;
        const float oc = C == 0 ? ol.x : (C == 1 ? ol.y : ol.z);
        const float oa = A == 0 ? ol.x : (A == 1 ? ol.y : ol.z), ob = B == 0 ? ol.x : (B == 1 ? ol.y : ol.z);
        const float num = __uint_as_float(q2.y) - oc;
        const float t = FAST ? div_by(by_dc, num) : num / dc;               // Hitables.cpp:95
        const float a = oa + t * da;
        const float b = ob + t * db;
        const bool in_t = (t >= t_min) & ((t < t_max) | ((t == t_max) & (idx > best)));
        const bool in_ab = (a >= __uint_as_float(q1.x)) & (a <= __uint_as_float(q1.y)) & (b >= __uint_as_float(q1.z)) & (b <= __uint_as_float(q2.x));
        const bool hit = in_t & in_ab;
        t_max = hit ? t : t_max;
        best = hit ? idx : best;
    }
}
// The list is walked in class order (mirt.hip sorts a copy at upload: the three unrotated rect orientations, unrotated
// spheres, everything else), each class by its own loop with the axes fixed at compile time and nothing to classify
// per primitive.  Walking out of order is safe because acceptance carries the scene index:
// "t <= t_max, a later primitive wins a tie" == "t < t_max || (t == t_max && index > best)" whatever the order.
template <bool PLDS, bool CHAIN = false>
__device__ __forceinline__ int32_t closest_list(const SceneDev &sc, const LdsView &v, f3 o, f3 d,
                                                float t_min, float &t_max)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    int32_t best = -1;
    uint32_t p = 0;
    ListPtr q = (ListPtr)sc.lprims;
    // The scene's flags and class ends are wave-uniform and loop-invariant; left to itself the compiler turns every test of them
    // into a 64-bit lane mask ahead of the kernel's main loop, spills the masks to VGPR lanes (the scalar file is full) and reads
    // each back with two v_readlane where it is tested -- ten vector instructions per pass.  Pinned here, they are plain 32-bit
    // scalars tested by s_bitcmp / s_cmp at the point of use.
    const uint32_t list_fast = sgpr_pin(sc.list_fast);
    const uint32_t lend0 = sgpr_pin(sc.lend[0]), lend1 = sgpr_pin(sc.lend[1]), lend2 = sgpr_pin(sc.lend[2]);
    if (lend2 != 0u) {
        // May the rect loops share their divisions (rect_loop<FAST>, div_by)?  Every lane's three direction components in
        // [2^-20, 2^20] and |k - (origin.c - offset.c)| <= 2^40: the upload vouches for |k|, |offset| <= 2^38 (list_fast), the
        // lane for its origin.  A NaN component passes the max3/min3 tests, harmlessly: both forms then give t = NaN.
        const float d_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
        const float d_lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
        const float o_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
        // (the scene's and the caller's part of the condition is wave-uniform and decided on the scalar unit first: and-ed into the
        // per-lane flag it was turned into lane masks and back -- eight vector instructions per pass)
        bool fast_wave = false;
        if (((list_fast & 1u) != 0u) && t_min >= 0x1p-19f) {
            // one ballot per compare (a ballot of a compare is that compare; a ballot of their conjunction costs a select and a compare more)
            fast_wave = (__builtin_amdgcn_ballot_w64(!(d_hi <= 0x1p20f)) | __builtin_amdgcn_ballot_w64(!(d_lo >= 0x1p-20f)) | __builtin_amdgcn_ballot_w64(!(o_hi <= 0x1p38f))) == 0ull;
        }
        if (fast_wave) {
            const Prims<PLDS> P(sc, v);
            rect_loop<0, 1, 2, true, PLDS>(q, p, lend0, o, d, t_min, t_max, best, list_fast, &P);
            rect_loop<0, 2, 1, true, PLDS>(q, p, lend1, o, d, t_min, t_max, best, list_fast, &P);
            rect_loop<2, 1, 0, true, PLDS>(q, p, lend2, o, d, t_min, t_max, best, list_fast, &P);
        } else {
            rect_loop<0, 1, 2, false>(q, p, lend0, o, d, t_min, t_max, best);
            rect_loop<0, 2, 1, false>(q, p, lend1, o, d, t_min, t_max, best);
            rect_loop<2, 1, 0, false>(q, p, lend2, o, d, t_min, t_max, best);
        }
    }

    const float a_dd = dot(d, d), two_a = 2.0f * a_dd, four_a = 4 * a_dd;   // Hitables.cpp:22,26,31 (unrotated spheres)
    float p_b = 0.0f, p_disc = 0.0f; int32_t p_idx = -1;                  // the parked sphere: b, discriminant, index (-1: none)
    // Which root survives (the first unless it is below t_min, Hitables.cpp:31-41) is decided by an estimate of the
    // first quotient wherever that is safe -- the estimate is within 3e-7 of the quotient, the band around t_min is
    // 1e-5 wide -- so one IEEE divide runs instead of two; inside the band (or with a NaN) both run as written.
    const float inv2a = __builtin_amdgcn_rcpf(two_a), t_lo = t_min * 0.99999f, t_hi = t_min * 1.00001f;
    auto resolve = [&]() {
        if (p_idx >= 0) {
            const float sq = MIRT_SQRT(p_disc);
            const float n1 = -p_b - sq, n2 = -p_b + sq;
            const float q1 = n1 * inv2a;
            const bool first = q1 > t_hi, second = q1 < t_lo;
            float t = (first ? n1 : n2) / two_a;
            if (!(first | second)) {
                t = n1 / two_a;                                          // Hitables.cpp:31
                if (!(t >= t_min)) t = n2 / two_a;                       // :41
            }
            if ((t >= t_min) & ((t < t_max) | ((t == t_max) & (p_idx > best)))) { t_max = t; best = p_idx; }
            p_idx = -1;
        }
    };
    for (; p < sc.lend[3]; ++p, q += 4) {
        const u32x4 q0 = q[0], q1 = q[1];
        const f3 oc = o - mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const float b = 2.0f * dot(d, oc);
        const float c = dot(oc, oc) - __uint_as_float(q1.z);
        const float bb = b * b;
        const float disc = bb - four_a * c;
        // A sphere wholly behind the origin (b > 0, c > 0) has two non-positive roots and is not parked.  disc < bb
        // makes that exact in float arithmetic: sqrt rounds monotonically, so sqrt(disc) <= b and -b + sqrt(disc) <= 0;
        // when four_a * c was absorbed (disc == bb) the sphere is parked and judged by its roots as written.
        const bool behind = (b > 0.0f) & (c > 0.0f) & (disc < bb);
        const bool cross = (disc > 0) & !behind;
        // "some lane with a parked sphere crosses this one", from one ballot per compare (each is that compare's own mask; the ballot
        // of the combined flag costs a select and a compare more per sphere)
        const unsigned long long m_cross = __builtin_amdgcn_ballot_w64(disc > 0) &
                                           ~(__builtin_amdgcn_ballot_w64(b > 0.0f) & __builtin_amdgcn_ballot_w64(c > 0.0f) & __builtin_amdgcn_ballot_w64(disc < bb));
        if (!(MIRT_TEXP & 32) && (m_cross & __builtin_amdgcn_ballot_w64(p_idx >= 0)) != 0ull) resolve();   // TEXP 32 (timing only): a second crossing overwrites the parked sphere
        if (cross) { p_b = b; p_disc = disc; p_idx = (int32_t)q1.w; }
    }
    resolve();
    for (; p < sc.n_prims; ++p, q += 4) {
        // every lane walks the same primitive: the record comes through the scalar cache straight into SGPRs (s_load;
        // the scene is read-only for the whole launch, so the constant address space is legitimate).
        const u32x4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        PrimRec pr;
        pr.off[0] = __uint_as_float(q0.x); pr.off[1] = __uint_as_float(q0.y); pr.off[2] = __uint_as_float(q0.z); pr.info = q0.w;
        pr.ga[0] = __uint_as_float(q1.x); pr.ga[1] = __uint_as_float(q1.y); pr.ga[2] = __uint_as_float(q1.z); pr.mat = q1.w;
        pr.gb[0] = __uint_as_float(q2.x); pr.gb[1] = __uint_as_float(q2.y); pr.rot[0] = __uint_as_float(q2.z); pr.rot[1] = __uint_as_float(q2.w);
        pr.rot[2] = __uint_as_float(q3.x); pr.rot[3] = __uint_as_float(q3.y); pr.rot[4] = __uint_as_float(q3.z); pr.rot[5] = __uint_as_float(q3.w);
        float t;
        if (prim_t<CHAIN>(pr, o, d, t_min, t_max, t, sc.ops) && (t < t_max || (int32_t)pr.mat > best)) { t_max = t; best = (int32_t)pr.mat; }
    }
    return best;
}

// AABB::Hit's arithmetic with its control flow folded away, result unchanged:
//  * (t0 > t_min) ? t0 : t_min is fmaxf(t0, t_min) and (t1 < t_max) ? t1 : t_max is fminf(t1, t_max): t_min/t_max are
//    never NaN, and a NaN t0/t1 (0 * inf) is ignored by both forms;
//  * t_min only grows and t_max only shrinks from axis to axis, so AABB.cpp:17's per-axis "t_max <= t_min -> miss"
//    fires on some axis exactly when it fires on the final values: one compare instead of three early returns.
// LEAF: the child is an object, which the reference never box-tests (SimpleObjectBVHNode::Hit tests its own box, then
// calls Object::Hit): its box may only cull what the primitives inside cannot hit, so a grazing interval
// (t_max == t_min: a ray through a box edge or corner, where a primitive can lie) passes.
__device__ __forceinline__ bool slab_hit(const float mn[3], const float mx[3], f3 o, f3 invd, float t_min, float t_max, uint32_t leaf, float &t_near)
{
#pragma unroll
    for (uint32_t a = 0; a < 3; ++a) {
        float id = get(invd, a);
        float t0 = (mn[a] - get(o, a)) * id;
        float t1 = (mx[a] - get(o, a)) * id;
        const bool neg = id < 0.0f;
        const float tn = neg ? t1 : t0, tf = neg ? t0 : t1;
        t_min = __builtin_fmaxf(tn, t_min);
        // the caller's t_max enters through a bare v_min_f32: the builtin first canonicalises an operand it cannot trace to an
        // arithmetic result (v_max_f32 x, x -- quieting a signalling NaN t_max never is), one instruction in every node step
        if (a == 0) asm("v_min_f32 %0, %1, %2" : "=v"(t_max) : "v"(tf), "v"(t_max));
        else t_max = __builtin_fminf(tf, t_max);
    }
    t_near = t_min;
    // leaf = 1 (the child is an object; Node2::lleaf / rleaf): "t_max < t_min -> miss", written as the node rule on the next float above t_max (t_max > 0
    // whenever the answer can be "hit"; +inf steps to NaN, which the negated compare passes)
    const float t_top = __uint_as_float(__float_as_uint(t_max) + leaf);
    return !(t_top <= t_min);
}

// BVH: SimpleObjectBVHNode::Hit (SimpleObject.cpp:389-406) as an explicit depth-first walk in the
// recursion's order: box test with the *current* t_max when the node is reached (AABB.cpp:5-21),
// left subtree completely, then right.  The pending right children live on a per-lane stack in
// LDS ([level][lane] so a level is one conflict-free row); nodes with index < lds_nodes (the top
// levels, nodes are numbered breadth-first at upload) are read from LDS, deeper ones from HBM as
// one 32-byte record per lane.
template <bool PLDS, bool CHAIN = false>
__device__ __forceinline__ int32_t closest_bvh(const SceneDev &sc, const LdsView &v, f3 o, f3 d,
                                               float t_min, float &t_max)
{
    int32_t best = -1;
    Stack stk(sc, v);
    int32_t cur = sc.root;
    const Prims<PLDS> P(sc, v);
    const f3 invd = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);              // AABB.cpp:9
    for (;;) {
        if (cur >= 0) {
            NodeRec n;
            if ((uint32_t)cur < sc.lds_nodes) n = ((const NodeRec *)v.nodes)[cur];
            else {
                const float4 *g = (const float4 *)(sc.nodes + cur);
                float4 a = g[0], b = g[1];
                n.mn[0] = a.x; n.mn[1] = a.y; n.mn[2] = a.z; n.left = __float_as_int(a.w);
                n.mx[0] = b.x; n.mx[1] = b.y; n.mx[2] = b.z; n.right = __float_as_int(b.w);
            }
            float t_in;
            if (slab_hit(n.mn, n.mx, o, invd, t_min, t_max, 0, t_in)) {   // AABB::Hit on this node's own box (never an object's)
                if (n.right != MIRT_CHILD_NONE) stk.push(n.right);
                cur = n.left;
                continue;
            }
        } else if (cur != MIRT_CHILD_NONE) {
            uint2 ob = sc.objs[~cur];                       // Object::Hit, SimpleObject.cpp:73-83
            for (uint32_t i = 0; i < ob.y; ++i) {
                float t;
                uint32_t p = ob.x + i;
                if (prim_t<CHAIN>(P[p], o, d, t_min, t_max, t, sc.ops)) { t_max = t; best = (int32_t)p; }
            }
        }
        if (stk.sp == 0) break;
        cur = stk.pop();
    }
    return best;
}

// The lanes of the wave for which `p` holds (s_and of the compare mask with exec; HIP's __ballot takes an int and costs a
// select and a compare more)
__device__ __forceinline__ unsigned long long wave_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ... and a pointer so pinned, which the compiler can no longer trace to a kernel argument, is told to be global memory
// again (it would otherwise be read with flat loads)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x4_t *GlobalF4;
typedef const __attribute__((address_space(3))) f32x4_t *LdsF4;
typedef const __attribute__((address_space(1))) u32x2_t *GlobalU2;

// An object's primitive tested from a BVH leaf.  Left to itself the compiler sinks every field's load into the branch that uses it
// (the sphere branch fetched r*r only after the kind had arrived and been tested: two memory round trips per leaf pass), so the
// first 32 bytes -- all a sphere needs, one sector -- are fetched together and pinned in front of the test; what only rects and
// rotated primitives need stays lazy.  Same arithmetic as prim_t on the record in memory.
template <bool PLDS, bool CHAIN = false>
__device__ __forceinline__ bool leaf_prim_t(const Prims<PLDS> &P, uint32_t p, f3 o, f3 d, float t_min, float t_max, float &t, const OpRec *ops = nullptr)
{
    if constexpr (PLDS) return prim_t<CHAIN>(P[p], o, d, t_min, t_max, t, ops);
    else {
        const GlobalF4 g = (GlobalF4)&P[p];
        f32x4_t r0 = g[0], r1 = g[1];
        asm volatile("" : "+v"(r0), "+v"(r1));
        PrimRec pr;
        pr.off[0] = r0.x; pr.off[1] = r0.y; pr.off[2] = r0.z; pr.info = __float_as_uint(r0.w);
        pr.ga[0] = r1.x; pr.ga[1] = r1.y; pr.ga[2] = r1.z; pr.mat = __float_as_uint(r1.w);
        // gb and rot are read by prim_t under the same two flags only
        if (pr.info & 1u) { pr.gb[0] = P[p].gb[0]; pr.gb[1] = P[p].gb[1]; }        // rects only (the second sector of the record)
        if (pr.info & 0x100u) {                                                   // rotation active
#pragma unroll
            for (int k = 0; k < 6; ++k) pr.rot[k] = P[p].rot[k];
        }
        if (CHAIN && (pr.info & 0x200u)) pr.rot[0] = P[p].rot[0];                 // chain form: where its ops start
        return prim_t<CHAIN>(pr, o, d, t_min, t_max, t, ops);
    }
}

// The leaf test of a scene whose objects are all single unrotated spheres (SceneDev::sph4): one 16-byte access per leaf instead of
// the record's first sector in two, and prim_t with the kind known at compile time (no dispatch, no rotation check) -- the same
// sphere arithmetic on the same operands.
__device__ __forceinline__ bool sphere_leaf_t(const float4 *sph4, uint32_t p, f3 o, f3 d, float t_min, float t_max, float &t, const DivBy &by2a, bool div_ok)
{
    const f32x4_t r = *(GlobalF4)((const __attribute__((address_space(1))) unsigned char *)sph4 + (p << 4));   // 32-bit byte offset (fewer than 2^26 objects, as nodes)
    PrimRec pr;
    pr.off[0] = r.x; pr.off[1] = r.y; pr.off[2] = r.z; pr.info = 0u /* MIRT_PRIM_SPHERE, no rotation */; pr.ga[2] = r.w;
    return prim_t(pr, o, d, t_min, t_max, t, nullptr, &by2a, div_ok);
}
// A walk's per-ray division constants for its sphere leaves (prim_t: by2a, div_ok)
struct LeafDiv {
    DivBy by2a; bool ok;
    __device__ __forceinline__ LeafDiv(f3 d, float t_min)
    {
        const float two_a = 2.0f * dot(d, d);
        by2a = div_prepare(two_a);
        if (!div_den_ok(two_a)) by2a.r1 = __builtin_nanf("");
        ok = MIRT_LEAF_DIVBY && t_min >= 0x1p-19f;
    }
};

// MIRT_NODE_SOA (measurement only, tools: profiles/r03_trace_ablation.txt): the near-first walk's nodes as sixteen planes of
// one dword per node ("coalesced SoA node fetches", north_star kernel item 2) instead of one 64-byte record.  A wave's lanes
// stand on unrelated nodes, so a plane read coalesces nothing: it is sixteen 4-byte accesses per lane where the record is four
// of 16 bytes.  Same tree, same arithmetic, same image.
#ifndef MIRT_NODE_SOA
#define MIRT_NODE_SOA 0
#endif
__device__ __forceinline__ void fetch_node2(const Node2 *nodes2, uint32_t n_nodes, int32_t cur, f32x4_t &a, f32x4_t &b, f32x4_t &c, f32x4_t &e)
{
#if MIRT_NODE_SOA
    const __attribute__((address_space(1))) float *pl = (const __attribute__((address_space(1))) float *)nodes2 + (cur >> 6);
    const size_t n = n_nodes;
    a.x = pl[0]; a.y = pl[n]; a.z = pl[2 * n]; a.w = pl[3 * n];
    b.x = pl[4 * n]; b.y = pl[5 * n]; b.z = pl[6 * n]; b.w = pl[7 * n];
    c.x = pl[8 * n]; c.y = pl[9 * n]; c.z = pl[10 * n]; c.w = pl[11 * n];
    e.x = pl[12 * n]; e.y = pl[13 * n]; e.z = pl[14 * n]; e.w = pl[15 * n];
#else
    // `cur` IS the byte offset (Node2::left; the upload refuses trees of 2^25 nodes and more): base in SGPRs + a 32-bit lane
    // offset costs nothing per step, where the signed 64-bit index arithmetic of `nodes2 + index` is a 64-bit shift and add
    const GlobalF4 g = (GlobalF4)((const __attribute__((address_space(1))) unsigned char *)nodes2 + (uint32_t)cur);
    a = g[0]; b = g[1]; c = g[2]; e = g[3];
#endif
}

// Near-first traversal of the same tree (ACCEL 2, 3, 4).  A node carries both children's boxes; each is tested
// with AABB::Hit's arithmetic (AABB.cpp:5-21; 1/dir is hoisted, it does not depend on the box) against the
// current t_max, the nearer child (larger-or-equal entry t goes on the stack) is visited first so t_max
// shrinks early and most of the far subtrees are culled by their box.  The closest t is the same as with
// the reference's left-first order; only the winner of an exact tie in t can differ.
// leaf_want: the loop is wave-synchronous -- one node step per pass for every lane that stands on a node -- and a lane that
// reaches a leaf (an object) WAITS there until `leaf_want` lanes of the wave wait with it, or no lane has a node left to
// step through; then all waiting lanes test their objects in one pass.  Testing objects whenever any lane stood on one
// (leaf_want = 1) ran that code -- the sphere test with its square root and divides -- in nearly every pass for the one lane
// in seven that needed it.  A waiting lane does nothing in between, so the sequence of box tests, object tests and t_max
// values of every ray is the same for any leaf_want: no result depends on it, nor on which rays share a wave.
// The walk is a resumable slice (option walk_yield): walks of one wave differ 2-3x in length, and a wave that waits for its
// longest walk leaves most lanes idle.  The wave leaves the walk loop as soon as `want` of the lanes that entered have
// finished; those are shaded (and start their next ray) while the others keep their walk -- node, stack pointer, t_max,
// best, 1/dir -- for the next slice.  The per-lane stack lives in LDS/HBM slots indexed by lane, so it survives between
// slices by itself.  want > 64 never yields: the plain walk (closest_bvh2 below).  Per-path arithmetic is the same either way.
// (Round 3 rewrote the step without divergent control flow -- child choice by selects on float keys, the push an unconditional
// LDS write, the stack's two newest entries fetched at the head of every pass so that pops are selects: bit-identical, SALU
// -9 %, lane use 54.5 -> 59.8 %, but VALU +10 % because every lane now pays for every case in every pass, and the walk is
// bound by VALU issue: 6-12 % SLOWER.  SALU instructions issue beside other waves' VALU work, so the branches this form
// removes cost little.  Leaf groups -- a node over at most four spheres collapsed into one leaf, its spheres consecutive 16-byte
// records tested in a loop -- were built too, with the oracle's twin: 11 % fewer node steps per ray but 1.8x the sphere tests, and
// those run in the leaf passes, where few lanes take part: 5 % slower on 485 spheres, 1.5 % on 1 M.  Numbers in
// profiles/r03_trace_ablation.txt; this is round 2's form.)
template <bool PLDS, bool CHAIN = false>
__device__ __forceinline__ int32_t closest_bvh2(const SceneDev &sc, const LdsView &v, f3 o, f3 d, float t_min, float &t_max, uint32_t leaf_want)
{
    int32_t best = -1;
    Stack stk(sc, v);
    // A finished walk is cur == MIRT_CHILD_NONE (the root of an empty tree already is): the three questions the loop asks the
    // wave -- who stands on a node, who on an object, who is still walking -- are then one compare of `cur` each, and a
    // ballot of a plain compare is that compare (a ballot of a flag carried around the loop costs a select and a compare more).
    int32_t cur = sc.root;
    const Prims<PLDS> P(sc, v);
    const Node2 *const nodes2 = sgpr_pin(sc.nodes2);
    const uint32_t n_nodes = MIRT_NODE_SOA ? sgpr_pin(sc.n_nodes) : 0u;
    const uint2 *const objs = sgpr_pin(sc.objs);
    const uint32_t single = sgpr_pin(sc.single);
    const float4 *const sph4 = sgpr_pin(sc.sph4);
    leaf_want = sgpr_pin(leaf_want);
    const f3 invd = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);              // AABB.cpp:9
    const LeafDiv ld(d, t_min);
    for (;;) {
        if (cur >= 0) {
            Node2 n;                                                      // always from memory: staging the top levels in LDS measured neutral, and the
            f32x4_t a, b, c, e;                                           // per-lane choice between two address spaces costs flat loads
            fetch_node2(nodes2, n_nodes, cur, a, b, c, e);
            n.lmn[0] = a.x; n.lmn[1] = a.y; n.lmn[2] = a.z; n.left = __float_as_int(a.w);
            n.lmx[0] = b.x; n.lmx[1] = b.y; n.lmx[2] = b.z; n.right = __float_as_int(b.w);
            n.rmn[0] = c.x; n.rmn[1] = c.y; n.rmn[2] = c.z; n.rmx[0] = e.x; n.rmx[1] = e.y; n.rmx[2] = e.z;
            float tl = 0.0f, tr = 0.0f;
            const bool hl = slab_hit(n.lmn, n.lmx, o, invd, t_min, t_max, __float_as_uint(c.w), tl);
            const bool hr = (n.right != MIRT_CHILD_NONE) & slab_hit(n.rmn, n.rmx, o, invd, t_min, t_max, __float_as_uint(e.w), tr);
            if (hl && hr) {
                const bool left_first = tl <= tr;
                stk.push(left_first ? n.right : n.left);
                cur = left_first ? n.left : n.right;
            } else if (hl) cur = n.left;
            else if (hr) cur = n.right;
            else if (stk.sp == 0) cur = MIRT_CHILD_NONE;
            else cur = stk.pop();
        }
        const bool at_leaf = (uint32_t)cur > (uint32_t)MIRT_CHILD_NONE;  // negative and not the end mark: ~object
        const unsigned long long lm = wave_mask(at_leaf);
        if (lm != 0ull && ((uint32_t)__popcll(lm) >= leaf_want || wave_mask(cur >= 0) == 0ull)) {
            if (at_leaf) {
                if (sph4) {                                               // wave-uniform: every object is one unrotated sphere
                    float t;
                    if (sphere_leaf_t(sph4, (uint32_t)~cur, o, d, t_min, t_max, t, ld.by2a, ld.ok)) { t_max = t; best = ~cur; }
                } else {
                    uint32_t first = (uint32_t)~cur, cnt = 1;
                    if (!single) { const u32x2_t ob = ((GlobalU2)objs)[~cur]; first = ob.x; cnt = ob.y; }      // Object::Hit, SimpleObject.cpp:73-83
                    for (uint32_t i = 0; i < cnt; ++i) {
                        float t;
                        if (leaf_prim_t<PLDS, CHAIN>(P, first + i, o, d, t_min, t_max, t, sc.ops)) { t_max = t; best = (int32_t)(first + i); }
                    }
                }
                if (stk.sp == 0) cur = MIRT_CHILD_NONE; else cur = stk.pop();
            }
        }
        if (wave_mask(cur != MIRT_CHILD_NONE) == 0ull) break;
    }
    return best;
}

struct WalkState { int32_t cur, best; float t_max; uint32_t sp; f3 invd; };

// LTOP (ACCEL 4): the first SceneDev::lds_nodes nodes (breadth-first numbering: the top levels) are read from the workgroup's LDS
// copy under an explicit per-lane branch -- ds_read, not a flat load, so those steps make no L1 access at all.  Pays where the walk
// waits on memory (1 M spheres +3.5 % with 32 nodes; 8/16/64: +1.5/+2.5/+2 %, 128 and more lose to the LDS they take); the branch
// costs the 485-sphere walk, which is issue-bound, 1-2 %, hence a variant of its own.
template <bool PLDS, bool LTOP, bool CHAIN = false>
__device__ __forceinline__ void bvh2_slice(const SceneDev &sc, const LdsView &v, f3 o, f3 d, float t_min, WalkState &w, uint32_t want, uint32_t leaf_want)
{
    // a lane takes part while w.cur != MIRT_CHILD_NONE (closest_bvh2's end mark); lanes that hold no walk enter with the mark
    Stack stk(sc, v);
    stk.sp = w.sp;
    int32_t cur = w.cur;
    const Prims<PLDS> P(sc, v);
    const f3 invd = w.invd;
    const Node2 *const nodes2 = sgpr_pin(sc.nodes2);
    const uint32_t n_nodes = MIRT_NODE_SOA ? sgpr_pin(sc.n_nodes) : 0u;
    const uint2 *const objs = sgpr_pin(sc.objs);
    const uint32_t single = sgpr_pin(sc.single);
    const float4 *const sph4 = sgpr_pin(sc.sph4);
    const uint32_t lds_n = LTOP ? sgpr_pin(sc.lds_nodes << 6) : 0u;   // in bytes, like cur
    leaf_want = sgpr_pin(leaf_want); want = sgpr_pin(want);
    const unsigned long long entered = wave_mask(cur != MIRT_CHILD_NONE);
    const LeafDiv ld(d, t_min);
    for (;;) {
        if (cur >= 0) {
            Node2 n;
            f32x4_t a, b, c, e;
            if (LTOP && (uint32_t)cur < lds_n) { const LdsF4 l = (LdsF4)(v.nodes + (uint32_t)cur); a = l[0]; b = l[1]; c = l[2]; e = l[3]; }   // the top of the tree: LDS reads, not L1 accesses
            else fetch_node2(nodes2, n_nodes, cur, a, b, c, e);
            n.lmn[0] = a.x; n.lmn[1] = a.y; n.lmn[2] = a.z; n.left = __float_as_int(a.w);
            n.lmx[0] = b.x; n.lmx[1] = b.y; n.lmx[2] = b.z; n.right = __float_as_int(b.w);
            n.rmn[0] = c.x; n.rmn[1] = c.y; n.rmn[2] = c.z; n.rmx[0] = e.x; n.rmx[1] = e.y; n.rmx[2] = e.z;
            float tl = 0.0f, tr = 0.0f;
            const bool hl = slab_hit(n.lmn, n.lmx, o, invd, t_min, w.t_max, __float_as_uint(c.w), tl);
            const bool hr = (n.right != MIRT_CHILD_NONE) & slab_hit(n.rmn, n.rmx, o, invd, t_min, w.t_max, __float_as_uint(e.w), tr);
            if (hl && hr) {
                const bool left_first = tl <= tr;
                stk.push(left_first ? n.right : n.left);
                cur = left_first ? n.left : n.right;
            } else if (hl) cur = n.left;
            else if (hr) cur = n.right;
            else if (stk.sp == 0) cur = MIRT_CHILD_NONE;
            else cur = stk.pop();
        }
        // lanes on a leaf wait for company (closest_bvh2); a slice may end while they wait: the leaf is then the lane's `cur`
        const bool at_leaf = (uint32_t)cur > (uint32_t)MIRT_CHILD_NONE;
        const unsigned long long lm = wave_mask(at_leaf);
        if (lm != 0ull && ((uint32_t)__popcll(lm) >= leaf_want || wave_mask(cur >= 0) == 0ull)) {
            if (at_leaf) {
                if (sph4) {                                               // wave-uniform: every object is one unrotated sphere
                    float t;
                    if (sphere_leaf_t(sph4, (uint32_t)~cur, o, d, t_min, w.t_max, t, ld.by2a, ld.ok)) { w.t_max = t; w.best = ~cur; }
                } else {
                    uint32_t first = (uint32_t)~cur, cnt = 1;
                    if (!single) { const u32x2_t ob = ((GlobalU2)objs)[~cur]; first = ob.x; cnt = ob.y; }      // Object::Hit, SimpleObject.cpp:73-83
                    for (uint32_t i = 0; i < cnt; ++i) {
                        float t;
                        if (leaf_prim_t<PLDS, CHAIN>(P, first + i, o, d, t_min, w.t_max, t, sc.ops)) { w.t_max = t; w.best = (int32_t)(first + i); }
                    }
                }
                if (stk.sp == 0) cur = MIRT_CHILD_NONE; else cur = stk.pop();
            }
        }
        const unsigned long long still = wave_mask(cur != MIRT_CHILD_NONE);
        if (still == 0ull || (uint32_t)__popcll(entered & ~still) >= want) break;
    }
    w.cur = cur; w.sp = stk.sp;
}

// Path ids enumerate pixels tile by tile (8x8) rather than row by row, so the 64 camera rays a wave generates at
// once cover a compact patch: their walks through a BVH stay together longer.  A bijection on the image (both sides
// multiples of 8; tiles_x = 0 turns it off); the pixel it returns is the real one -- the Philox key, the camera ray,
// the slot of the sum -- so no result depends on it.  k_accumulate applies the same map.
__device__ __forceinline__ uint32_t tile_pixel(uint32_t g, uint32_t tiles_x, uint32_t width)
{
    if (tiles_x == 0u) return g;
    const uint32_t tile = g >> 6, in = g & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    return (ty * 8u + (in >> 3)) * width + tx * 8u + (in & 7u);
}
__device__ __forceinline__ uint32_t tile_pixel(uint32_t g, const FastDiv &by_tiles_x, uint32_t width)   // by_tiles_x.d == tiles_x
{
    if (by_tiles_x.d == 0u) return g;
    const uint32_t tile = g >> 6, in = g & 63u;
    const uint32_t ty = fast_div(tile, by_tiles_x), tx = tile - ty * by_tiles_x.d;
    return (ty * 8u + (in >> 3)) * width + tx * 8u + (in & 7u);
}

// ---------------------------------------------------------------------------------- ray generation
// HomemadeRayTracer.cpp:114-131 + SimpleCamera.cpp:55-60 + Randomizer.h:50-58
template <class Args>
__device__ __forceinline__ void generate_ray(const Args &a, uint32_t pixel, uint32_t sample, f3 &o, f3 &d)
{
    const uint32_t j = fast_div(pixel, a.by_width), i = pixel - j * a.width;
    PhiloxKeyed g; g.pixel = pixel; g.sample = sample; g.stage = 0; g.k0 = a.k0; g.k1 = a.k1;
    uint32_t r[4]; g.block(0, r);
    float u, v;
    if (a.flags & MIRT_FLAG_ONE_SPP) {
        u = (float)i / (float)a.width;
        v = (float)j / (float)a.height;
    } else {
        u = ((float)i + map_unorm(r[0])) / (float)a.width;
        v = ((float)j + map_unorm(r[1])) / (float)a.height;
    }
    // RandomInUnitDisk: iteration 0 = block 0 lanes 2,3; iterations 2k-1, 2k = block k.  With a pinhole
    // (lens radius 0, wave-uniform) the disk point is multiplied by 0 (SimpleCamera.cpp:57) and the keyed
    // stream has no state to advance, so the loop is skipped; offset is then +-0 either way.
    f3 p = mk(0.0f, 0.0f, 0.0f);
    if (a.cam.lens_radius != 0.0f) {
        p = mk(map_pm1(r[2]), map_pm1(r[3]), 0.0f);
        for (uint32_t it = 1; it < MIRT_REJECT_CAP && dot(p, p) >= 1.0f; ++it) {
            uint32_t k = 2u + 2u * it;
            if ((k & 3u) == 0u) g.block(k >> 2, r);
            p = mk(map_pm1(r[k & 3u]), map_pm1(r[(k & 3u) + 1u]), 0.0f);
        }
    }
    camera_ray(a.cam, u, v, p, o, d);
}

// Set bits of a wave mask below this lane (v_mbcnt_lo + v_mbcnt_hi: two instructions, no lane mask to build)
__device__ __forceinline__ uint32_t rank_in(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---------------------------------------------------------------------------------- wave-cooperative draws
// Rejection sampling of the unit sphere (Randomizer.h:40-48) accepts 52 % of its candidates, so a wave
// whose lanes each loop "until mine is accepted" runs ~6 rounds with a third of its lanes working.
// The draws are keyed (pixel, sample, stage, iteration) and any lane can evaluate any key, so after
// round 0 the still-pending lanes publish their keys in LDS and ALL 64 lanes evaluate candidates for
// them: with n pending, each gets G = pow2 <= 64/n consecutive iterations in one round.  The owner
// takes the first accepted iteration, exactly the one its own loop would have stopped at.
// SHARE > 0: the owner publishes the head of its Philox blocks (rt_device.h philox_head: the rounds that are the same for
// every iteration, which it has from its own block 0) instead of the key, and the helpers finish the block for their
// iteration: 2, 3 or 4 of a block's 20 multiplications less per candidate, for four or five head words kept by every lane
// through the rounds.  The list kernels take it (MIRT_LIST_SHARE); the BVH kernels, compiled for 64 registers, spill on it.
// `it` in bytes (iterations stay below MIRT_REJECT_CAP + 8): the list kernel's seven workgroups per CU leave ~1.6 KB of LDS a
// workgroup for the scene's primitives and materials, which the two-box Cornell scene needs.
template <int SHARE> struct CoopLdsT { uint32_t w[SHARE == 0 ? 3 : (SHARE == 3 ? 5 : 4)][64]; uint8_t it[64]; };
static_assert(MIRT_REJECT_CAP + 8 <= 256, "CoopLdsT::it is a byte");
typedef CoopLdsT<0> CoopLds;

// Is the candidate (the first three words of a Philox block) inside the unit sphere?  The reference decides on
// the mapped floats (Randomizer.h:43-46: reject while dot(p,p) >= 1), and mapping three words costs 21 f64
// instructions.  Here the squared length is first taken exactly in integers: a word r maps to 2r/(2^32-1) - 1, which is
// (r - 2^31) / 2^31 to within 5e-10, so S = sum (r_i - 2^31)^2 is the squared length in units of 2^-62 (three 32x32+64
// multiply-adds, no conversions).  The mapped floats' dot product is within 2e-6 of it: outside a band of 8e-6 around 1
// the integer decides, inside it (about one candidate in 10^5) the reference's own arithmetic does.  Only the accepted
// candidate is mapped exactly, once, by its owner.
// candidate_score: the squared length in units of 2^-30 with the band already decided (0 / ONE inside it), so that "inside" is
// the plain compare score < CAND_ONE -- which a wave-wide ballot gets for nothing.
constexpr uint32_t CAND_ONE = 0x40000000u;
__device__ __forceinline__ uint32_t candidate_score(const uint32_t r[4])
{
    // sum (r_i - 2^31)^2 = sum r_i^2 - 2^32 * sum r_i + 3 * 2^62, evaluated modulo 2^64 (the true value is <= 3 * 2^62, so
    // nothing is lost): unsigned multiply-adds, and the other two terms only touch the high word
    const uint64_t Q = (uint64_t)r[0] * r[0] + (uint64_t)r[1] * r[1] + (uint64_t)r[2] * r[2];
    uint32_t hi = (uint32_t)(Q >> 32) - (r[0] + r[1] + r[2]) + 0xC0000000u;   // squared length in units of 2^-30
    constexpr uint32_t BAND = 8600u;                                         // 8600 / 2^30 = 8.0e-6
    if (hi - (CAND_ONE - BAND) < 2u * BAND) {
        const f3 p = mk(map_pm1(r[0]), map_pm1(r[1]), map_pm1(r[2]));
        hi = !(dot(p, p) >= 1.0f) ? 0u : CAND_ONE;
    }
    return hi;
}
__device__ __forceinline__ bool candidate_inside(const uint32_t r[4]) { return candidate_score(r) < CAND_ONE; }

// need: NEED_* of this lane (NEED_NONE for idle lanes; every lane of the wave must call this).
// LSTAGE: the stage (1 + depth) differs from lane to lane (k_trace) and travels with the key.
// ALL_EVAL: every lane finishes a block in a cooperative round, needed or not (see there); without it only the helpers do.
template <bool LSTAGE, int SHARE = 0, bool ALL_EVAL = false>
__device__ __forceinline__ void coop_draws(CoopLdsT<SHARE> &cl, uint32_t lane, uint32_t need, uint32_t pixel, uint32_t sample,
                                           uint32_t stage, uint32_t k0, uint32_t k1, f3 &sphere_pt, float &unorm)
{
    constexpr int HS = SHARE == 0 ? 1 : SHARE;                  // (PhiloxHead<0> does not exist)
    // Wave-wide questions are asked as ballots of plain compares (a ballot of a compare is the compare's own mask; a ballot of a
    // flag carried around the loop costs a select and a compare more): "pending" is nit != 0, "accepted" a compare of the score.
    uint32_t nit = 0u;                                          // a pending lane's next iteration (>= 1); 0: not pending
    // The accepted candidate's raw words (set in round 0, read only by lanes that need a sphere point) and the Philox head (published
    // by pending lanes only) start as "whatever the register holds": a zero fill is nine moves per pass for values nobody reads.
    // (An empty asm defines them without an instruction; leaving them uninitialised would be undefined behaviour in the language.)
    uint32_t w0 = any_vgpr(), w1 = any_vgpr(), w2 = any_vgpr();
    PhiloxHead<HS> head;
#pragma unroll
    for (int i = 0; i < (HS >= 3 ? 5 : 4); ++i) head.w[i] = any_vgpr();
    if (need != NEED_NONE) {                                    // round 0: block 0 serves all three materials
        uint32_t r[4];
        if constexpr (SHARE != 0) { head = philox_head<HS>(pixel, sample, stage, k0, k1); philox_tail<HS>(head, 0u, k0, k1, r); }
        else philox4x32_10(pixel, sample, stage, 0, k0, k1, r);
        unorm = map_unorm(r[0]);
        w0 = r[0]; w1 = r[1]; w2 = r[2];                       // read only when need == NEED_SPHERE
        nit = ((need == NEED_SPHERE) & !candidate_inside(r) & !(MIRT_TEXP & 4)) ? 1u : 0u;
    }
    unsigned long long mask;
    while ((mask = __builtin_amdgcn_ballot_w64(nit != 0u)) != 0ull) {
        const bool pending = nit != 0u;
        // n and lg are wave-uniform: counted in two 32-bit halves they stay on the scalar unit (compares of the 64-bit population
        // count compiled to three v_cmp_*_u64 per round -- there is no scalar 64-bit less-than)
        const uint32_t n = (uint32_t)__builtin_popcount((uint32_t)mask) + (uint32_t)__builtin_popcount((uint32_t)(mask >> 32));
        // G = 1,2,4,8 iterations per pending lane for n > 32, > 16, > 8, <= 8: lg = 3 - clamp(ceil(log2 n) - 3, 0, 3), integer arithmetic only
        const uint32_t cl2 = 32u - (uint32_t)__builtin_clz((n - 1u) | 1u);       // ceil(log2 n) for n >= 2 (1 for n = 1; n >= 1 here)
        const uint32_t over = cl2 > 3u ? cl2 - 3u : 0u;
        const uint32_t lg = 3u - (over < 3u ? over : 3u);
        const uint32_t G = 1u << lg;
        const uint32_t rank = rank_in(mask);
        if (pending) {
            cl.it[rank] = (uint8_t)nit;
            if constexpr (SHARE != 0) {
#pragma unroll
                for (int i = 0; i < (SHARE == 3 ? 5 : 4); ++i) cl.w[i][rank] = head.w[i];
            } else { cl.w[0][rank] = pixel; cl.w[1][rank] = sample; if (LSTAGE) cl.w[2][rank] = stage; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t e = lane >> lg, sub = lane & (G - 1u);
        const uint32_t it = (uint32_t)cl.it[e] + sub;             // e < 64 always; entries >= n are stale and unused
        unsigned long long am;                                    // lanes whose candidate is accepted
        uint32_t r[4];
        if constexpr (ALL_EVAL) {
            // Every lane finishes a block -- a helper for its owner's iteration, the others (e >= n: stale entries, or beyond the cap)
            // for nobody: skipping them took a branch, a zero fill of r and a ballot of a flag, and saved nothing (the wave runs the
            // block anyway).  Whose candidate counts is decided in the masks.  (List kernel and the sliced walk of small trees:
            // +1.7 % / +1 %; the 1 M-sphere kernel, at its register cap, spills one more dword with it and loses 1.4 %.)
            if constexpr (SHARE != 0) {
                PhiloxHead<HS> hh;
#pragma unroll
                for (int i = 0; i < (SHARE == 3 ? 5 : 4); ++i) hh.w[i] = cl.w[i][e];
                philox_tail<HS>(hh, it, k0, k1, r);
            } else philox4x32_10(cl.w[0][e], cl.w[1][e], LSTAGE ? cl.w[2][e] : stage, it, k0, k1, r);
            const uint32_t score = candidate_score(r);
            // accepted: a helper (e < n) within the cap whose candidate is inside, or the cap's last candidate, kept like the per-lane loop keeps it
            am = __builtin_amdgcn_ballot_w64(e < n) & __builtin_amdgcn_ballot_w64(it < MIRT_REJECT_CAP) &
                 (__builtin_amdgcn_ballot_w64(score < CAND_ONE) | __builtin_amdgcn_ballot_w64(it == MIRT_REJECT_CAP - 1u));
        } else {
            r[0] = r[1] = r[2] = r[3] = 0u;
            bool acc = false;
            if ((e < n) & (it < MIRT_REJECT_CAP)) {
                if constexpr (SHARE != 0) {
                    PhiloxHead<HS> hh;
#pragma unroll
                    for (int i = 0; i < (SHARE == 3 ? 5 : 4); ++i) hh.w[i] = cl.w[i][e];
                    philox_tail<HS>(hh, it, k0, k1, r);
                } else philox4x32_10(cl.w[0][e], cl.w[1][e], LSTAGE ? cl.w[2][e] : stage, it, k0, k1, r);
                acc = candidate_inside(r) | (it == MIRT_REJECT_CAP - 1u);              // the cap keeps its last candidate, like the per-lane loop
            }
            am = __ballot(acc);
        }
        // a lane that is not pending can have rank == n, and n * G can be 64: its shift count is clamped (a 64-bit shift by 64 is undefined)
        const uint32_t sh = pending ? (rank << lg) : 0u;
        const uint32_t bits = (uint32_t)(am >> sh) & ((1u << G) - 1u);             // this lane's G helpers, lowest = earliest iteration
        const bool got = pending & (bits != 0u);
        // byte address of the helper lane for ds_bpermute (what __shfl computes, minus its width arithmetic); the find-first
        // of an empty set is used only under `got`
        const int src4 = (int)((got ? sh + (uint32_t)__builtin_ctz(bits) : lane) << 2);
        nit = (pending & !got) ? nit + G : 0u;
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)r[0]), c1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)r[1]),
                       c2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)r[2]);
        w0 = got ? c0 : w0; w1 = got ? c1 : w1; w2 = got ? c2 : w2;
        __builtin_amdgcn_wave_barrier();                          // the LDS keys are rewritten next round
    }
    if (need == NEED_SPHERE) sphere_pt = mk(map_pm1(w0), map_pm1(w1), map_pm1(w2));
}

// ---------------------------------------------------------------------------------- the bounce kernel
template <bool FIRST, int ACCEL /*0 list, 1 bvh walked in the reference's order, 2 bvh near-first*/, bool PLDS>
__global__ void __launch_bounds__(MIRT_BLOCK) k_bounce(BounceArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_wcnt[2][MIRT_BLOCK / 64];   // double-buffered by iteration parity: no third barrier
    __shared__ uint32_t s_base[2];
    __shared__ CoopLds s_coop[MIRT_BLOCK / 64];

    __shared__ uint32_t s_work;
    const uint32_t n_in = FIRST ? a.n_first : *a.res_in;
    uint32_t ush = a.chunk_shift;                                  // unit = 1 << ush slots, >= 4 units per workgroup if possible
    while (ush > 8u && (n_in >> ush) < gridDim.x * 4u) --ush;
    if (((uint64_t)blockIdx.x << ush) >= n_in) return;
    const LdsView lv = lds_view(a.sc, smem);
    lds_stage(a.sc, lv);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t chunk = 1u << a.chunk_shift;
    // thread 0's view of the workgroup's open out-chunk [ck_base, ck_base + chunk), filled up to ck_next
    uint32_t ck_base = 0xFFFFFFFFu, ck_next = 0, blk_alive = 0, par = 0;

    for (;;) {
      if (threadIdx.x == 0) s_work = atomicAdd(a.cursor, 1u);
      __syncthreads();
      const uint32_t ci = s_work;
      __syncthreads();                                                          // s_work is rewritten next round
      const uint64_t ubase = (uint64_t)ci << ush;
      if (ubase >= n_in) break;
      // the unit's paths end where its chunk's filled part ends (the tail of a chunk is unused)
      const uint64_t cbase = ubase & ~(uint64_t)(chunk - 1u);
      const uint64_t filled = FIRST ? (uint64_t)n_in : cbase + a.len_in[ubase >> a.chunk_shift];   // uniform
      const uint64_t uend = ubase + (1ull << ush);
      const uint32_t stop = (uint32_t)(filled < uend ? filled : uend);
      for (uint32_t base = (uint32_t)ubase; base < stop; base += MIRT_BLOCK) {
        const uint32_t idx = base + threadIdx.x;
        const bool active = idx < stop;
        f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1.0f, 1.0f, 1.0f);
        uint32_t pixel = 0, sample = 0;
        if (active) {
            if (FIRST) {
                uint32_t sl = idx / a.npix, pl = idx - sl * a.npix;
                pixel = a.pix0 + pl; sample = a.spp0 + sl;
                generate_ray(a, pixel, sample, o, d);
            } else {
                if (EXP_ON(32)) { o = mk(0.1f * lane, 0.0f, 0.5f); d = mk(0.3f, -0.5f + 0.01f * lane, -1.0f); pixel = idx % (a.npix); sample = a.spp0; }
                else {
                const float4 q0 = a.in.p0[idx], q1 = a.in.p1[idx], q2 = a.in.p2[idx];
                o = mk(q0.x, q0.y, q0.z); d = mk(q0.w, q1.x, q1.y); thr = mk(q1.z, q1.w, q2.x);
                pixel = __float_as_uint(q2.y); sample = __float_as_uint(q2.z);
                }
            }
        }
        // -------- phase 1 (per lane): extend, HomemadeRayTracer.cpp:160-162, and classify the vertex
        bool done = false;                    // path ends at this vertex with `radiance`
        f3 radiance = mk(0.0f, 0.0f, 0.0f);
        uint32_t need = NEED_NONE;
        Hit h; h.pos = o; h.nrm = d; h.mat = 0;
        const MatRec *M = a.sc.mats;
        if (active) {
            float t_max = 3.402823466e+38f;   // FLT_MAX
            const float t_min = 0.001f;
            int32_t best;
            if (EXP_ON(8)) { best = (int32_t)(lane % 9u); t_max = 1.0f + 0.01f * lane; }
            else best = ACCEL == 2 ? closest_bvh2<PLDS>(a.sc, lv, o, d, t_min, t_max, MIRT_LEAF_WANT)
                      : (ACCEL == 1 ? closest_bvh<PLDS>(a.sc, lv, o, d, t_min, t_max) : closest_list<PLDS>(a.sc, lv, o, d, t_min, t_max));
            if (best < 0) {                                                        // :186
                radiance = thr * mk(a.sc.ambient[0], a.sc.ambient[1], a.sc.ambient[2]); done = true;
            } else {
                const Prims<PLDS> P(a.sc, lv);
                prim_record(P[best], best, o, d, t_max, false, h);
                if (a.flags & MIRT_FLAG_NORMAL_DISPLAY) {                          // :164-167
                    radiance = scale(mk(h.nrm.x + 1.0f, h.nrm.y + 1.0f, h.nrm.z + 1.0f), 0.5f); done = true;
                } else if (!(a.depth < a.max_depth)) {                             // Scatter not called: emitted stays zero (:173-174)
                    done = true;
                } else {
                    M = ((uint32_t)h.mat < a.sc.lds_mats) ? (const MatRec *)(lv.mats + h.mat) : (a.sc.mats + h.mat);
                    need = scatter_need(M->kind, d, h.nrm, true);
                }
            }
        }
        // -------- phase 2 (whole wave): the draws of Scatter
        f3 sp = mk(0.0f, 0.0f, 0.0f); float un = 0.0f;
        if (EXP_ON(2)) { sp = mk(0.1f, 0.2f, 0.3f); un = 0.5f; }
        else coop_draws<false>(s_coop[wave], lane, need, pixel, sample, 1u + a.depth, a.k0, a.k1, sp, un);
        // -------- phase 3 (per lane): the rest of Scatter, forward (throughput) order of :176
        bool alive = false;
        f3 no = o, nd = d;
        if (active && !done) {
            Scat s;
            if (EXP_ON(4)) { s.alive = (lane & 7u) != 0u; s.dir = sp + h.nrm; s.atten = mk(0.5f, 0.5f, 0.5f); s.emitted = mk(0.0f, 0.0f, 0.0f); }
            else scatter_finish(*M, mk(M->albedo[0], M->albedo[1], M->albedo[2]), d, h.pos, h.nrm, sp, un, s);   // engine 0 renders single-colour scenes only
            if (s.alive) { alive = true; no = h.pos; nd = s.dir; thr = thr * s.atten; }
            else { radiance = thr * s.emitted; done = true; }
        }
        if (active && done && !EXP_ON(1)) {
            uint32_t slot = (sample - a.spp0) * a.npix + (pixel - a.pix0);
            store_rad(a.rad, slot, radiance);
        }
        // -------- compaction: wave ballot + prefix popcount; slots come from the workgroup's open chunk
        const unsigned long long m = __ballot(alive);
        const uint32_t wcnt = (uint32_t)__popcll(m);
        const uint32_t wpre = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[par][wave] = wcnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < MIRT_BLOCK / 64; ++w) tot += s_wcnt[par][w];
            uint32_t at = 0xFFFFFFFFu;
            if (tot) {
                if (ck_base == 0xFFFFFFFFu || ck_next + tot > ck_base + chunk) {
                    if (ck_base != 0xFFFFFFFFu) a.len_out[ck_base >> a.chunk_shift] = ck_next - ck_base;   // close
                    uint32_t nb = atomicAdd(a.res_out, chunk);
                    if (nb + chunk > a.cap_slots) { atomicOr(a.overflow, 1u); ck_base = 0xFFFFFFFFu; }
                    else { ck_base = nb; ck_next = nb; }
                }
                if (ck_base != 0xFFFFFFFFu) { at = ck_next; ck_next += tot; blk_alive += tot; }
            }
            s_base[par] = at;
        }
        __syncthreads();
        const uint32_t sb = s_base[par];
        if (EXP_ON(16)) asm volatile("" :: "v"(no.x), "v"(no.y), "v"(no.z), "v"(nd.x), "v"(nd.y), "v"(nd.z), "v"(thr.x), "v"(thr.y), "v"(thr.z));
        if (alive && sb != 0xFFFFFFFFu && !EXP_ON(16)) {
            uint32_t pos = sb + wpre;
            for (uint32_t w = 0; w < wave; ++w) pos += s_wcnt[par][w];
            a.out.p0[pos] = make_float4(no.x, no.y, no.z, nd.x);
            a.out.p1[pos] = make_float4(nd.y, nd.z, thr.x, thr.y);
            a.out.p2[pos] = make_float4(thr.z, __uint_as_float(pixel), __uint_as_float(sample), 0.0f);
        }
        par ^= 1u;
      }
    }
    if (threadIdx.x == 0) {
        if (ck_base != 0xFFFFFFFFu) a.len_out[ck_base >> a.chunk_shift] = ck_next - ck_base;
        if (blk_alive) atomicAdd(a.alive_out, blk_alive);
    }
}


// ---------------------------------------------------------------------------------- the persistent kernel
// k_trace: the same per-vertex work as k_bounce (extend, classify, cooperative draws, finish) without the
// path queues.  Each lane owns one path and walks it bounce after bounce; when the path ends, the lane
// takes a fresh camera ray.  Camera rays are produced 64 at a time by the whole wave (full lanes, coherent
// pixels) into a small per-wave LDS stash, so refilling a lane costs a few LDS reads; path ids come from a
// global cursor in claims of a few hundred, one atomic per claim.  Nothing but the scene and the 12-byte
// radiance slot per path touches HBM, and one launch serves a whole batch.
// Every path's arithmetic and Philox keys are those of k_bounce, so the two engines give identical images.
struct TraceArgs {
    RadSlot *rad;
    uint32_t *cursor;              // next unclaimed path id of the batch
    uint32_t *hist;                // [0..63] paths ended at depth d (63: at 63 or deeper), [64] vertices beyond depth 63
    uint32_t n_paths, claim;       // claim: path ids taken per atomic (multiple of 64)
    // Every wave of the grid first works through chunks of 64 path ids that are its own without asking: chunk k of wave w is
    // ids [(k * grid_waves + w) * 64, +64) for k < own_chunks -- dealt round-robin, so that each wave's share is spread over
    // the whole batch (contiguous shares are contiguous image rows, and rows differ in work).  Only the ids from
    // first_claim = grid_waves * own_chunks * 64 on are claimed from the cursor, which balances the end of the launch.
    // One atomic word hands out ~90 M claims a second; a launch that claimed everything in units of 64 was bound by it.
    uint32_t own_chunks, grid_waves, first_claim;
    SceneDev sc;
    mirt_camera cam;
    uint32_t width, height, pix0, npix, spp0;
    uint32_t max_depth, flags, k0, k1;
    uint32_t tiles_x;              // > 0: path ids run over 8x8 pixel tiles (tile_pixel), tiles_x = width / 8
    FastDiv by_npix, by_width, by_tiles_x;   // reciprocals of the three divisors of a path id (rt_device.h fast_div), made by the host
    uint32_t slice_want;           // ACCEL 3: finished walks that end a slice (bvh2_slice)
    uint32_t leaf_want;            // ACCEL 2, 3: lanes waiting on a leaf that start an object-test pass (closest_bvh2)
    // second stream (MIRT_DEFER_MD, list walk): per-wave queues in HBM, plane-major.  mdq: MDQ_PLANES float4 planes of MDQ_CAP
    // parked metal/dielectric vertices; rdy: RDY_PLANES planes of RDY_CAP rays ready to be traced.  NULL: not used.
    float4 *mdq, *rdy;
};
#ifndef MIRT_DEFER_MD
#define MIRT_DEFER_MD 0
#endif
#define MDQ_CAP 128u
#define MDQ_PLANES 5u
#define RDY_CAP 256u
#define RDY_PLANES 3u
// 128 entries: topped up whenever fewer than 64 remain, so one pass always serves every idle lane (list walk:
// the cheapest loop).  64 entries: drained completely before the refill, two passes when the idle lanes
// outnumber what is left (BVH walks: iterations are long, LDS is what limits residency).
template <int N, bool PIX = false>
struct StashLds { float4 a[N], b[N]; uint32_t pix[PIX ? N : 1]; };   // a = (ox, oy, oz, dx), b = (dy, dz, path id, sample of the batch): two 16-byte LDS accesses per ray;
                                                                      // pix (BVH kernels, whose path ids are tiled): the ray's pixel, which its generation computed anyway

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FULL: the kernel variants for scenes with image textures (hits carry u,v -- CalculateUV's atan2f/asinf for spheres -- and the
// albedo is fetched) and/or primitives in chain form (general instance chains, rt_device.h); plain scenes compile neither
#ifndef MIRT_LIST_STASH
#define MIRT_LIST_STASH 128u
#endif
#ifndef MIRT_LIST_SHARE
#define MIRT_LIST_SHARE 2         // Philox rounds the helpers of a cooperative draw take from the owner (coop_draws): list kernel
#endif
#ifndef MIRT_BVH_SHARE
#define MIRT_BVH_SHARE 0          // ... BVH kernels (spills under their 64-register cap: profiles/r03_trace_ablation.txt)
#endif
#ifndef MIRT_LIST_WAVES
#define MIRT_LIST_WAVES 7
#endif
#ifndef MIRT_YIELD_WAVES
#define MIRT_YIELD_WAVES 8   // the yielding walk keeps its walk state live through the shading code: 3-4 dwords of scratch under the 64-register cap
#endif
// The kernel's arguments as they lie in the kernarg segment, behind a pointer the compiler cannot see through: what the
// rare blocks of the loop read (camera, image geometry: ~40 words that only the stash refill needs) is then fetched there
// with scalar loads instead of living in SGPRs across the whole loop -- where it did not fit, and was spilled to VGPR lanes
// and read back with v_readlane in the hot path.  TraceArgs is the kernels' only parameter (offset 0 of the segment).
__device__ __forceinline__ const TraceArgs &late_args()
{
    typedef const __attribute__((address_space(4))) unsigned char *KargPtr;
    KargPtr kp = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const TraceArgs *)kp;
}

template <int ACCEL, bool PLDS, bool FULL>
__device__ __forceinline__ void trace_body(const TraceArgs &a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int SHARE = ACCEL == 0 ? MIRT_LIST_SHARE : MIRT_BVH_SHARE;
    __shared__ CoopLdsT<SHARE> s_coop[MIRT_BLOCK / 64];
    constexpr bool BIG_STASH = ACCEL == 0;
    constexpr uint32_t STASH_N = BIG_STASH ? MIRT_LIST_STASH : 64u;
    __shared__ StashLds<STASH_N, ACCEL != 0> s_stash[MIRT_BLOCK / 64];
    __shared__ uint32_t s_hist[65];

    if (a.own_chunks == 0u && (uint64_t)blockIdx.x * a.claim >= a.n_paths) return;   // more workgroups than claims (uniform)
    const LdsView lv = lds_view(a.sc, smem);
    if (threadIdx.x < 65) s_hist[threadIdx.x] = 0;
    lds_stage(a.sc, lv);                                            // ends with a barrier
    // the wave's index is wave-uniform, and the compiler must be told: derived from threadIdx it counts as divergent, and with it
    // the whole of the wave's bookkeeping (claimed ids, stash count) -- vector registers and vector instructions for scalar work
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto &stash = s_stash[wave];

    // wave-uniform bookkeeping
    uint32_t cnt = 0;                       // rays in the stash
    uint32_t w_next = 0, w_end = 0;         // owned or claimed, not yet generated path ids
    uint32_t own_k = 0;                     // chunks of its own the wave has started
    uint32_t x_made = 0, x_iters = 0;       // MIRT_TEXP & 16: per-wave schedule dump (tools/wave_dump.py)
    const unsigned long long x_t0 = (MIRT_TEXP & 16) ? wall_clock64() : 0ull;
    const unsigned long long x_c0 = (MIRT_TEXP & 64) ? __builtin_readcyclecounter() : 0ull;   // MIRT_TEXP 16 + 64: the dump carries shader-clock cycles instead of the hardware ids
    // the next run of path ids for this wave (w_next == w_end): its next own chunk, else a claim; false when the batch is handed out
    auto next_range = [&]() -> bool {
        if (own_k < a.own_chunks) {
            w_next = (own_k * a.grid_waves + (blockIdx.x * (MIRT_BLOCK / 64) + wave)) * 64u; w_end = w_next + 64u;
            ++own_k;
            return true;
        }
        // (Guided self-scheduling -- claims that shrink towards the end of the batch -- was measured and lost: waves of one SIMD
        // advance at very different speeds, 0.3 to 2.7 paths/us within one launch (tools/wave_dump.py: the oldest wave is served
        // first), and the last exits of a 28.8 ms launch are spread over 1.2 ms; but shrinking to 64 ids cost 5 ms at the claim
        // counter, and floors of 256..1024 ids left the time where it was.)
        // (Dealing a workgroup's claim to its waves 64 ids at a time through a pool in LDS -- one global atomic per 256..512 ids
        // and still the fine grain -- was built and measured too: no faster than waves that keep their whole claim.)
        uint32_t b = 0;
        if (lane == 0) b = a.first_claim + atomicAdd(a.cursor, a.claim);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        if (b >= a.n_paths) return false;
        w_next = b; w_end = (a.n_paths - b < a.claim) ? a.n_paths : b + a.claim;
        return true;
    };
    // the pixel of a path id's pixel index: the list walk never tiles its path ids (mirt.hip sets tiles_x = 0 for it); the BVH
    // walks divide by the plain argument here (its reciprocal, three more SGPRs live across their loop, measured 2 % slower)
    auto path_pixel = [&](uint32_t g) -> uint32_t { if constexpr (ACCEL == 0) return g; else return tile_pixel(g, a.tiles_x, a.width); };
    bool exhausted = false;
    // the lane's path
    bool live = false;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1.0f, 1.0f, 1.0f);
    uint32_t pid = 0, pixel = 0, sample = 0, depth = 0;
    WalkState ws; ws.cur = MIRT_CHILD_NONE; ws.best = -1; ws.t_max = 0.0f; ws.sp = 0; ws.invd = mk(0, 0, 0);   // ACCEL 3 only; cur: no walk under way

    // ---- second stream: metal / dielectric vertices are parked (after their draws) and shaded 64 at a time
    constexpr bool DEFER = MIRT_DEFER_MD && ACCEL == 0 && !FULL;
    uint32_t q_cnt = 0, r_cnt = 0;                                   // wave-uniform: parked vertices, ready rays
    const uint32_t gwave = blockIdx.x * (MIRT_BLOCK / 64) + wave;
    float4 *const mdq = DEFER ? a.mdq + (size_t)gwave * MDQ_PLANES * MDQ_CAP : nullptr;
    float4 *const rdy = DEFER ? a.rdy + (size_t)gwave * RDY_PLANES * RDY_CAP : nullptr;
    auto service = [&](uint32_t n) {                                  // the top n parked vertices, one per lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // the parking stores of earlier passes have landed ...
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");         // ... before other lanes read them
        bool cont = false, dead = false;
        f3 so = mk(0, 0, 0), sd = mk(0, 0, 0), sthr = mk(0, 0, 0);
        uint32_t spid = 0, sdepth = 0, ssl = 0;
        if (lane < n) {
            const uint32_t e = q_cnt - n + lane;
            const float4 p0 = mdq[e], p1 = mdq[MDQ_CAP + e], p2 = mdq[2u * MDQ_CAP + e], p3 = mdq[3u * MDQ_CAP + e], p4 = mdq[4u * MDQ_CAP + e];
            const f3 pos = mk(p0.x, p0.y, p0.z), din = mk(p0.w, p1.x, p1.y), nrm = mk(p1.z, p1.w, p2.x), spt = mk(p2.y, p2.z, p2.w);
            sthr = mk(p3.x, p3.y, p3.z); spid = __float_as_uint(p3.w);
            sdepth = __float_as_uint(p4.x); ssl = __float_as_uint(p4.z);
            const uint32_t smi = __float_as_uint(p4.y);
            const MatRec *SM = (PLDS || smi < a.sc.lds_mats) ? (const MatRec *)((const unsigned char *)lv.mats + __umul24(smi, (uint32_t)sizeof(MatRec))) : (a.sc.mats + smi);
            Scat s;
            scatter_finish<false, 2>(*SM, mk(SM->albedo[0], SM->albedo[1], SM->albedo[2]), din, pos, nrm, spt, spt.x, s);
            if (s.alive) { cont = true; so = pos; sd = s.dir; sthr = sthr * s.atten; }
            else dead = true;
        }
        if (dead) {                                                   // absorbed by the metal: thr * (0,0,0), :173
            const f3 z = scale(sthr, 0.0f);
            store_rad(a.rad, spid, z);
            atomicAdd(&s_hist[sdepth < 63u ? sdepth : 63u], 1u);
            if (sdepth > 63u) atomicAdd(&s_hist[64], sdepth - 63u);
        }
        const unsigned long long cm = __ballot(cont);
        if (cont) {
            const uint32_t e = r_cnt + rank_in(cm);
            rdy[e] = make_float4(so.x, so.y, so.z, sd.x);
            rdy[RDY_CAP + e] = make_float4(sd.y, sd.z, sthr.x, sthr.y);
            rdy[2u * RDY_CAP + e] = make_float4(sthr.z, __uint_as_float(spid), __uint_as_float(sdepth + 1u), __uint_as_float(ssl));
        }
        r_cnt += (uint32_t)__popcll(cm);
        q_cnt -= n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };

    for (;;) {
        if constexpr (DEFER) {
            if (q_cnt >= 64u && r_cnt + 64u <= RDY_CAP) service(64u);
            // idle lanes take ready rays before fresh camera rays
            const unsigned long long idle0 = __ballot(!live);
            if (idle0 != 0ull && r_cnt != 0u) {
                const uint32_t n_idle = (uint32_t)__popcll(idle0);
                const uint32_t take = n_idle < r_cnt ? n_idle : r_cnt;
                const uint32_t rank = rank_in(idle0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (!live && rank < take) {
                    const uint32_t e = r_cnt - 1u - rank;
                    const float4 q0 = rdy[e], q1 = rdy[RDY_CAP + e], q2 = rdy[2u * RDY_CAP + e];
                    o = mk(q0.x, q0.y, q0.z); d = mk(q0.w, q1.x, q1.y); thr = mk(q1.z, q1.w, q2.x);
                    pid = __float_as_uint(q2.y); depth = __float_as_uint(q2.z);
                    const uint32_t sl = __float_as_uint(q2.w);
                    sample = a.spp0 + sl; pixel = path_pixel(a.pix0 + (pid - sl * a.npix));
                    live = true;
                }
                r_cnt -= take;
            }
        }
        if constexpr (BIG_STASH) {
            // ---- top up the stash: 64 camera rays by the whole wave (HomemadeRayTracer.cpp:114-131)
            if (cnt + 64u <= STASH_N && !exhausted) {
                if (w_next == w_end && !next_range()) exhausted = true;
                if (!exhausted) {
                    const uint32_t g = (w_end - w_next < 64u) ? w_end - w_next : 64u;
                    if (lane < g) {
                        const TraceArgs &ka = late_args();
                        const uint32_t id = w_next + lane;
                        const uint32_t sl = fast_div(id, ka.by_npix), pl = id - sl * ka.npix;
                        f3 go, gd;
                        generate_ray(ka, tile_pixel(ka.pix0 + pl, ka.by_tiles_x, ka.width), ka.spp0 + sl, go, gd);
                        const uint32_t e = cnt + lane;
                        stash.a[e] = make_float4(go.x, go.y, go.z, gd.x);
                        stash.b[e] = make_float4(gd.y, gd.z, __uint_as_float(id), __uint_as_float(sl));
                    }
                    w_next += g; cnt += g; x_made += g;
                    wave_lds_sync();
                }
            }
            // ---- idle lanes take a fresh path from the top of the stash
            const unsigned long long idle = __ballot(!live);
            if (idle != 0ull && cnt != 0u) {
                const uint32_t rank = rank_in(idle);
                if (!live && rank < cnt) {
                    const uint32_t e = cnt - 1u - rank;
                    const float4 qa = stash.a[e], qb = stash.b[e];
                    o = mk(qa.x, qa.y, qa.z); d = mk(qa.w, qb.x, qb.y); pid = __float_as_uint(qb.z);
                    const uint32_t sl = __float_as_uint(qb.w);
                    sample = a.spp0 + sl; pixel = path_pixel(a.pix0 + (pid - sl * a.npix));
                    thr = mk(1.0f, 1.0f, 1.0f); depth = 0; live = true;
                }
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                cnt -= sgpr_pin((n_idle < cnt) ? n_idle : cnt);            // (pinned: left alone the pair becomes a saturating subtraction, which only the vector unit has)
                wave_lds_sync();                                        // the freed entries are rewritten by the next top-up
            }
        } else {
            // ---- idle lanes take what the stash holds; if lanes are still idle, the whole wave makes 64 new
            // camera rays (HomemadeRayTracer.cpp:114-131) into the now empty stash and serves the rest
            const unsigned long long idle = __ballot(!live);
            if (idle != 0ull && (cnt != 0u || !exhausted)) {
                // without a loop: what the stash holds, then at most one round of 64 new rays (a claim that ends short of the idle
                // lanes leaves the rest idle for one pass).  Written as a loop ("until every idle lane is served") the compiler took
                // the wave's bookkeeping -- counts, claimed ids -- for divergent and kept it in vector registers
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                const uint32_t rank = rank_in(idle);
                auto take_from = [&](uint32_t first, uint32_t take) {
                    if (!live && rank >= first && rank < first + take) {
                        const uint32_t e = cnt - 1u - (rank - first);
                        const float4 qa = stash.a[e], qb = stash.b[e];
                        o = mk(qa.x, qa.y, qa.z); d = mk(qa.w, qb.x, qb.y); pid = __float_as_uint(qb.z);
                        const uint32_t sl = __float_as_uint(qb.w);
                        sample = a.spp0 + sl;
                        if constexpr (ACCEL != 0) pixel = stash.pix[e]; else pixel = path_pixel(a.pix0 + (pid - sl * a.npix));
                        thr = mk(1.0f, 1.0f, 1.0f); depth = 0; live = true;
                    }
                };
                const uint32_t take1 = n_idle < cnt ? n_idle : cnt;
                if (take1 != 0u) { take_from(0u, take1); cnt -= take1; wave_lds_sync(); }
                if (take1 != n_idle && !exhausted) {                  // the stash is empty now
                    if (w_next == w_end && !next_range()) exhausted = true;
                    else {
                        const uint32_t g = (w_end - w_next < 64u) ? w_end - w_next : 64u;
                        if (lane < g) {
                            const TraceArgs &ka = late_args();
                            const uint32_t id = w_next + lane;
                            const uint32_t sl = fast_div(id, ka.by_npix), pl = id - sl * ka.npix;
                            f3 go, gd;
                            const uint32_t px = tile_pixel(ka.pix0 + pl, ka.by_tiles_x, ka.width);
                            generate_ray(ka, px, ka.spp0 + sl, go, gd);
                            stash.a[lane] = make_float4(go.x, go.y, go.z, gd.x);
                            stash.b[lane] = make_float4(gd.y, gd.z, __uint_as_float(id), __uint_as_float(sl));
                            if constexpr (ACCEL != 0) stash.pix[lane] = px;
                        }
                        w_next += g; cnt = g; x_made += g;
                        wave_lds_sync();
                        const uint32_t take2 = (n_idle - take1 < cnt) ? n_idle - take1 : cnt;
                        take_from(take1, take2); cnt -= take2;
                        wave_lds_sync();
                    }
                }
            }
        }
        const unsigned long long live_now = __ballot(live);
        if (MIRT_TEXP & 16) ++x_iters;
        if (live_now == 0ull) {                                     // pool exhausted, stash empty, every path ended
            if constexpr (DEFER) { if (q_cnt != 0u) { service(q_cnt < 64u ? q_cnt : 64u); continue; } if (r_cnt != 0u) continue; }
            break;
        }
        if constexpr (DEFER) {
            // end of the launch (the refill above could not fill the wave): parked vertices need not wait for company any more
            if (q_cnt != 0u && (uint32_t)__popcll(live_now) <= 16u && r_cnt + 64u <= RDY_CAP) service(q_cnt < 64u ? q_cnt : 64u);
        }

        // ---- phase 1 (per lane): extend, HomemadeRayTracer.cpp:160-162, and classify the vertex
        bool done = false;
        f3 radiance;                           // set wherever `done` is
        uint32_t need = NEED_NONE;
        Hit h;                                 // set where `need`/`done` are decided below; read only under `live && !done`
        const MatRec *M;
        if constexpr (ACCEL != 0) { h.pos = o; h.nrm = d; h.mat = 0; M = a.sc.mats; radiance = mk(0.0f, 0.0f, 0.0f); }   // under the 64-register cap of the BVH kernels defined values allocate better (fewer spills)
        bool ready = live;                      // lanes whose closest hit is known in this iteration
        if constexpr (ACCEL >= 3) {
            if (live) {
                if (ws.cur == MIRT_CHILD_NONE) {                                   // a new ray: start its walk (a walk under way keeps its state)
                    ws.cur = a.sc.root; ws.best = -1; ws.t_max = 3.402823466e+38f; ws.sp = 0;
                    ws.invd = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);              // AABB.cpp:9
                }
                bvh2_slice<PLDS, ACCEL == 4, FULL>(a.sc, lv, o, d, 0.001f, ws, a.slice_want, a.leaf_want);
            }
            ready = live && ws.cur == MIRT_CHILD_NONE;
        }
        if (ready) {
            float t_max = 3.402823466e+38f;   // FLT_MAX
            const float t_min = 0.001f;
            int32_t best;
            if constexpr (ACCEL >= 3) { best = ws.best; t_max = ws.t_max; }
            else best = ACCEL == 2 ? closest_bvh2<PLDS, FULL>(a.sc, lv, o, d, t_min, t_max, a.leaf_want)
                      : (ACCEL == 1 ? closest_bvh<PLDS, FULL>(a.sc, lv, o, d, t_min, t_max) : closest_list<PLDS, FULL>(a.sc, lv, o, d, t_min, t_max));
            if (best < 0) {                                                        // :186
                radiance = thr * mk(a.sc.ambient[0], a.sc.ambient[1], a.sc.ambient[2]); done = true;
            } else {
                const Prims<PLDS> P(a.sc, lv);
                prim_record<FULL>(P[best], best, o, d, t_max, FULL && a.sc.n_textures != 0u, h, a.sc.ops);
                if (a.flags & MIRT_FLAG_NORMAL_DISPLAY) {                          // :164-167
                    radiance = scale(mk(h.nrm.x + 1.0f, h.nrm.y + 1.0f, h.nrm.z + 1.0f), 0.5f); done = true;
                } else if (!(depth < a.max_depth)) {                               // :173-174
                    radiance = mk(0.0f, 0.0f, 0.0f); done = true;
                } else {
                    // PLDS launches have the whole material table in LDS as well (mirt.hip): a plain LDS pointer, not a
                    // per-lane choice between two address spaces (flat loads and 64-bit address arithmetic)
                    if constexpr (PLDS) M = (const MatRec *)((const unsigned char *)lv.mats + __umul24((uint32_t)h.mat, (uint32_t)sizeof(MatRec)));
                    else M = a.sc.mats + h.mat;                    // always from memory here: a per-lane choice of address space would make every
                                                                   // material read a flat load (the texture addresser's work even when it ends in LDS)
                    need = scatter_need(M->kind, d, h.nrm, true);
                }
            }
        }
        // ---- phase 2 (whole wave): the draws of Scatter, keyed by each lane's own depth
        f3 sp = mk(0.0f, 0.0f, 0.0f); float un = 0.0f;
        coop_draws<true, SHARE, ACCEL == 0 || ACCEL == 3>(s_coop[wave], lane, need, pixel, sample, 1u + depth, a.k0, a.k1, sp, un);
        // ---- phase 3 (per lane): the rest of Scatter, forward (throughput) order of :176
        bool parked = false, retry = false;
        if constexpr (DEFER) {
            // metal / dielectric vertices leave the lane: their state goes to the wave's queue and the lane is free for another path
            const bool md = ready && !done && ((M->kind & 2u) != 0u);             // MIRT_MAT_METAL = 2, MIRT_MAT_DIELECTRIC = 3
            const unsigned long long mm = __ballot(md);
            if (mm != 0ull) {
                const uint32_t e = q_cnt + rank_in(mm);
                if (md) {
                    if (e < MDQ_CAP) {
                        const uint32_t mi = (uint32_t)h.mat;
                        const float sx = need == NEED_UNORM ? un : sp.x;
                        mdq[e] = make_float4(h.pos.x, h.pos.y, h.pos.z, d.x);
                        mdq[MDQ_CAP + e] = make_float4(d.y, d.z, h.nrm.x, h.nrm.y);
                        mdq[2u * MDQ_CAP + e] = make_float4(h.nrm.z, sx, sp.y, sp.z);
                        mdq[3u * MDQ_CAP + e] = make_float4(thr.x, thr.y, thr.z, __uint_as_float(pid));
                        mdq[4u * MDQ_CAP + e] = make_float4(__uint_as_float(depth), __uint_as_float(mi), __uint_as_float(sample - a.spp0), 0.0f);
                        parked = true;
                    } else retry = true;                                         // queue full: the lane keeps its ray and meets this vertex again
                }
                const uint32_t n_md = (uint32_t)__popcll(mm);
                q_cnt = (q_cnt + n_md < MDQ_CAP) ? q_cnt + n_md : MDQ_CAP;                // the stores are awaited where they are read (service)
            }
        }
        if (ready && !done && !parked && !retry) {
            Scat s;
            const f3 albedo = (FULL && a.sc.n_textures != 0u) ? albedo_of(*M, a.sc.texs, a.sc.texels, h.u, h.v) : mk(M->albedo[0], M->albedo[1], M->albedo[2]);
            scatter_finish<ACCEL != 0, DEFER ? 1 : 0>(*M, albedo, d, h.pos, h.nrm, sp, un, s, ACCEL == 0 ? (int)(need == NEED_SPHERE) : -1);   // (the BVH kernels, at their register cap, take the dot product again)
            if (s.alive) { o = h.pos; d = s.dir; thr = thr * s.atten; }
            else { radiance = s.emits ? thr * s.emitted : scale(thr, 0.0f); done = true; }   // an absorbed path: thr * (0,0,0), :173
        }
        if (parked) live = false;
        if (ready && done) {
            if (!(MIRT_TEXP & 8)) store_rad(a.rad, pid, radiance);
            atomicAdd(&s_hist[depth < 63u ? depth : 63u], 1u);
            if (depth > 63u) atomicAdd(&s_hist[64], depth - 63u);
            live = false;
        }
        if (ready && !retry) ++depth;
    }
    if ((MIRT_TEXP & 16) && a.mdq && lane == 0) {
        const unsigned long long t1 = wall_clock64();
        const uint32_t gw = blockIdx.x * (MIRT_BLOCK / 64) + wave;
        const unsigned long long x_c = (MIRT_TEXP & 64) ? __builtin_readcyclecounter() - x_c0 : 0ull;
        if (MIRT_TEXP & 64) a.mdq[2 * gw] = make_float4(__uint_as_float(x_made), __uint_as_float(x_iters), __uint_as_float((uint32_t)x_c), __uint_as_float((uint32_t)(x_c >> 32)));
        else
        a.mdq[2 * gw] = make_float4(__uint_as_float(x_made), __uint_as_float(x_iters), __uint_as_float(__builtin_amdgcn_s_getreg(63492)), __uint_as_float(__builtin_amdgcn_s_getreg(63508)));
        a.mdq[2 * gw + 1] = make_float4(__uint_as_float((uint32_t)x_t0), __uint_as_float((uint32_t)(x_t0 >> 32)), __uint_as_float((uint32_t)t1), __uint_as_float((uint32_t)(t1 >> 32)));
    }
    __syncthreads();
    if (threadIdx.x < 65 && s_hist[threadIdx.x]) atomicAdd(a.hist + threadIdx.x, s_hist[threadIdx.x]);
}

// ---------------------------------------------------------------------------------- two paths per lane (list walk)
// k_trace2 (option paths_per_lane = 2; measured, NOT the default): the persistent kernel with TWO paths in every lane.  A wave
// pass costs its instructions whether 9 or 64 lanes need them, and the metal/dielectric half of Scatter is ~160 instructions
// for ~9 lanes.  With two independent paths per lane that pass serves 128 paths: a lane shades whichever of its two paths
// needs it (a lane whose two paths both do keeps the second for the next iteration, where it meets the same vertex and the
// same keyed draws again); the pending draws of both paths share their cooperative rounds, in-loop sphere roots share a pass,
// and every scene record fetched by the list walk is tested against two rays.  Per-path arithmetic, Philox keys and the
// radiance slot are those of k_trace: the images are bit-identical (test_two_paths_per_lane_is_bit_identical).
// Measured on Cornell 1024^2@512 (profiles/r02_two_paths.txt): lane use 68.2 -> 72.6 %, but 887 VALU instructions per 64
// rays against 885 -- the selects that route two paths through one pass, the second owner's side of every cooperative round
// and the extra root passes cost what the shared pass saves -- and 95 VGPRs leave 5 waves per SIMD instead of 7, which issue
// 4 % fewer instructions per second: 58.0 ms against 54.7.  4 waves measure the same, 6 (80 VGPRs, spills) 66 ms.
#ifndef MIRT_POOL_DRAWS
#define MIRT_POOL_DRAWS 1
#endif
#ifndef MIRT_LIST2_WAVES
#define MIRT_LIST2_WAVES 5
#endif
template <int A, int B, int C, bool FAST>
__device__ __forceinline__ void rect_loop2(ListPtr &q, uint32_t &p, uint32_t end, const f3 (&o)[2], const f3 (&d)[2],
                                           float t_min, float (&t_max)[2], int32_t (&best)[2])
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if (p >= end) return;
    float dc[2], da[2], db[2]; DivBy by_dc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        dc[k] = C == 0 ? d[k].x : (C == 1 ? d[k].y : d[k].z);
        da[k] = A == 0 ? d[k].x : (A == 1 ? d[k].y : d[k].z); db[k] = B == 0 ? d[k].x : (B == 1 ? d[k].y : d[k].z);
        by_dc[k] = div_prepare(dc[k]);
    }
    for (; p < end; ++p, q += 4) {
        const u32x4 q0 = q[0], q1 = q[1];
        const u32x2 q2 = *(const __attribute__((address_space(4))) u32x2 *)(q + 2);
        const int32_t idx = (int32_t)q1.w;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                       // rect_loop's body, once per path
            const f3 ol = o[k] - mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
            const float oc = C == 0 ? ol.x : (C == 1 ? ol.y : ol.z);
            const float oa = A == 0 ? ol.x : (A == 1 ? ol.y : ol.z), ob = B == 0 ? ol.x : (B == 1 ? ol.y : ol.z);
            const float num = __uint_as_float(q2.y) - oc;
            const float t = FAST ? div_by(by_dc[k], num) : num / dc[k];     // Hitables.cpp:95
            const float a = oa + t * da[k];
            const float b = ob + t * db[k];
            const bool in_t = (t >= t_min) & ((t < t_max[k]) | ((t == t_max[k]) & (idx > best[k])));
            const bool in_ab = (a >= __uint_as_float(q1.x)) & (a <= __uint_as_float(q1.y)) & (b >= __uint_as_float(q1.z)) & (b <= __uint_as_float(q2.x));
            const bool hit = in_t & in_ab;
            t_max[k] = hit ? t : t_max[k];
            best[k] = hit ? idx : best[k];
        }
    }
}
// closest_list for the two rays of a lane: the same tests in the same order per ray, one record fetch for both
template <bool PLDS>
__device__ __forceinline__ void closest_list2(const SceneDev &sc, const LdsView &v, const f3 (&o)[2], const f3 (&d)[2],
                                              float t_min, float (&t_max)[2], int32_t (&best)[2])
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    best[0] = -1; best[1] = -1;
    uint32_t p = 0;
    ListPtr q = (ListPtr)sc.lprims;
    if (sc.lend[2] != 0u) {
        bool fast = ((sc.list_fast & 1u) != 0u) & (t_min >= 0x1p-19f);
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                        // closest_list's range test, both rays
            const float d_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d[k].x), __builtin_fabsf(d[k].y)), __builtin_fabsf(d[k].z));
            const float d_lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(d[k].x), __builtin_fabsf(d[k].y)), __builtin_fabsf(d[k].z));
            const float o_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o[k].x), __builtin_fabsf(o[k].y)), __builtin_fabsf(o[k].z));
            fast = fast & (d_hi <= 0x1p20f) & (d_lo >= 0x1p-20f) & (o_hi <= 0x1p38f);
        }
        if (__ballot(!fast) == 0ull) {
            rect_loop2<0, 1, 2, true>(q, p, sc.lend[0], o, d, t_min, t_max, best);
            rect_loop2<0, 2, 1, true>(q, p, sc.lend[1], o, d, t_min, t_max, best);
            rect_loop2<2, 1, 0, true>(q, p, sc.lend[2], o, d, t_min, t_max, best);
        } else {
            rect_loop2<0, 1, 2, false>(q, p, sc.lend[0], o, d, t_min, t_max, best);
            rect_loop2<0, 2, 1, false>(q, p, sc.lend[1], o, d, t_min, t_max, best);
            rect_loop2<2, 1, 0, false>(q, p, sc.lend[2], o, d, t_min, t_max, best);
        }
    }
    float two_a[2], four_a[2], inv2a[2], p_b[2] = { 0.0f, 0.0f }, p_disc[2] = { 0.0f, 0.0f }; int32_t p_idx[2] = { -1, -1 };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float a_dd = dot(d[k], d[k]);                                  // Hitables.cpp:22,26,31 (unrotated spheres)
        two_a[k] = 2.0f * a_dd; four_a[k] = 4 * a_dd; inv2a[k] = __builtin_amdgcn_rcpf(two_a[k]);
    }
    const float t_lo = t_min * 0.99999f, t_hi = t_min * 1.00001f;
    // The roots of parked spheres, one pass for both paths: a lane resolves the parked sphere of path 0 if `w0`, else that of
    // path 1 if `w1` (closest_list's `resolve`, operands selected).  Lanes that want both are served by a second pass.
    auto resolve_pass = [&](bool w0, bool w1) {
        const bool s1 = !w0;
        if (w0 | w1) {
            const float pb = s1 ? p_b[1] : p_b[0], pd = s1 ? p_disc[1] : p_disc[0], ta = s1 ? two_a[1] : two_a[0], ia = s1 ? inv2a[1] : inv2a[0];
            const float tm = s1 ? t_max[1] : t_max[0];
            const int32_t pi = s1 ? p_idx[1] : p_idx[0], bs = s1 ? best[1] : best[0];
            const float sq = MIRT_SQRT(pd);
            const float n1 = -pb - sq, n2 = -pb + sq;
            const float q1 = n1 * ia;
            const bool first = q1 > t_hi, second = q1 < t_lo;
            float t = (first ? n1 : n2) / ta;
            if (!(first | second)) {
                t = n1 / ta;                                                 // Hitables.cpp:31
                if (!(t >= t_min)) t = n2 / ta;                              // :41
            }
            const bool acc = (t >= t_min) & ((t < tm) | ((t == tm) & (pi > bs)));
            t_max[0] = (acc & !s1) ? t : t_max[0]; best[0] = (acc & !s1) ? pi : best[0]; p_idx[0] = s1 ? p_idx[0] : -1;
            t_max[1] = (acc & s1) ? t : t_max[1];  best[1] = (acc & s1) ? pi : best[1];  p_idx[1] = s1 ? -1 : p_idx[1];
        }
    };
    for (; p < sc.lend[3]; ++p, q += 4) {
        const u32x4 q0 = q[0], q1 = q[1];
        bool cross[2]; float nb[2], nd[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f3 oc = o[k] - mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
            const float b = 2.0f * dot(d[k], oc);
            const float c = dot(oc, oc) - __uint_as_float(q1.z);
            const float bb = b * b;
            const float disc = bb - four_a[k] * c;
            const bool behind = (b > 0.0f) & (c > 0.0f) & (disc < bb);       // see closest_list
            cross[k] = (disc > 0) & !behind;
            nb[k] = b; nd[k] = disc;
        }
        bool w0 = cross[0] & (p_idx[0] >= 0), w1 = cross[1] & (p_idx[1] >= 0);
        while (__ballot(w0 | w1) != 0ull) { resolve_pass(w0, w1); w1 = w0 & w1; w0 = false; }
#pragma unroll
        for (int k = 0; k < 2; ++k) if (cross[k]) { p_b[k] = nb[k]; p_disc[k] = nd[k]; p_idx[k] = (int32_t)q1.w; }
    }
    if (__ballot(p_idx[0] >= 0) != 0ull) resolve_pass(p_idx[0] >= 0, false);      // the end of the list: each path by itself (a lane with
    if (__ballot(p_idx[1] >= 0) != 0ull) resolve_pass(false, p_idx[1] >= 0);      // a sphere parked on both would force two passes anyway)
    for (; p < sc.n_prims; ++p, q += 4) {
        const u32x4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        PrimRec pr;
        pr.off[0] = __uint_as_float(q0.x); pr.off[1] = __uint_as_float(q0.y); pr.off[2] = __uint_as_float(q0.z); pr.info = q0.w;
        pr.ga[0] = __uint_as_float(q1.x); pr.ga[1] = __uint_as_float(q1.y); pr.ga[2] = __uint_as_float(q1.z); pr.mat = q1.w;
        pr.gb[0] = __uint_as_float(q2.x); pr.gb[1] = __uint_as_float(q2.y); pr.rot[0] = __uint_as_float(q2.z); pr.rot[1] = __uint_as_float(q2.w);
        pr.rot[2] = __uint_as_float(q3.x); pr.rot[3] = __uint_as_float(q3.y); pr.rot[4] = __uint_as_float(q3.z); pr.rot[5] = __uint_as_float(q3.w);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float t;
            if (prim_t(pr, o[k], d[k], t_min, t_max[k], t) && (t < t_max[k] || (int32_t)pr.mat > best[k])) { t_max[k] = t; best[k] = (int32_t)pr.mat; }
        }
    }
}

// coop_draws for the two paths of a lane: round 0 per path, then the pending draws of BOTH paths share the cooperative rounds
// (up to 128 keys in LDS; a round serves the first 64 / G of them, the others wait their turn).  The owner of a key still takes
// its first accepted iteration, so every path gets the very candidate its own loop would have stopped at.
struct CoopLds2 { uint32_t pix[128], smp[128], it[128], stg[128]; };
__device__ __forceinline__ void coop_draws2(CoopLds2 &cl, uint32_t lane, const uint32_t (&need)[2], const uint32_t (&pixel)[2], const uint32_t (&sample)[2],
                                            const uint32_t (&stage)[2], uint32_t k0, uint32_t k1, f3 (&sphere_pt)[2], float (&unorm)[2])
{
    bool pending[2] = { false, false };
    uint32_t next_it[2] = { 1u, 1u };
    uint32_t w[2][3] = { { 0u, 0u, 0u }, { 0u, 0u, 0u } };         // the accepted candidates' raw words
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (need[k] != NEED_NONE) {                                 // round 0: block 0 serves all three materials
            uint32_t r[4];
            philox4x32_10(pixel[k], sample[k], stage[k], 0, k0, k1, r);
            unorm[k] = map_unorm(r[0]);
            w[k][0] = r[0]; w[k][1] = r[1]; w[k][2] = r[2];
            pending[k] = (need[k] == NEED_SPHERE) & !candidate_inside(r);
        }
    }
    for (;;) {
        const unsigned long long m0 = __ballot(pending[0]), m1 = __ballot(pending[1]);
        if ((m0 | m1) == 0ull) break;
        const uint32_t n0 = (uint32_t)__popcll(m0), n = n0 + (uint32_t)__popcll(m1);
        const uint32_t lg = n > 32 ? 0u : (n > 16 ? 1u : (n > 8 ? 2u : 3u));     // G = 1,2,4,8 iterations per served key
        const uint32_t G = 1u << lg, n_served = n < (64u >> lg) ? n : (64u >> lg);
        const uint32_t rank[2] = { rank_in(m0), n0 + rank_in(m1) };
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (pending[k]) { cl.pix[rank[k]] = pixel[k]; cl.smp[rank[k]] = sample[k]; cl.it[rank[k]] = next_it[k]; cl.stg[rank[k]] = stage[k]; }
        wave_lds_sync();
        const uint32_t e = lane >> lg, sub = lane & (G - 1u);
        uint32_t r[4] = { 0u, 0u, 0u, 0u };
        bool acc = false;
        const uint32_t it = cl.it[e] + sub;                       // e < 64; entries >= n are stale and unused
        if ((e < n_served) & (it < MIRT_REJECT_CAP)) {
            philox4x32_10(cl.pix[e], cl.smp[e], cl.stg[e], it, k0, k1, r);
            acc = candidate_inside(r) | (it == MIRT_REJECT_CAP - 1u);              // the cap keeps its last candidate, like the per-lane loop
        }
        const unsigned long long am = __ballot(acc);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool served = pending[k] & (rank[k] < n_served);
            const uint32_t sh = served ? (rank[k] << lg) : 0u;                        // < 64 for a served key
            const uint32_t bits = (uint32_t)(am >> sh) & ((1u << G) - 1u);            // its G helpers, lowest = earliest iteration
            const bool got = served & (bits != 0u);
            const int src4 = (int)((got ? sh + (uint32_t)__builtin_ctz(bits) : lane) << 2);
            next_it[k] += (served & !got) ? G : 0u;
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)r[0]), c1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)r[1]),
                           c2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)r[2]);
            w[k][0] = got ? c0 : w[k][0]; w[k][1] = got ? c1 : w[k][1]; w[k][2] = got ? c2 : w[k][2];
            pending[k] = pending[k] & !got;
        }
        __builtin_amdgcn_wave_barrier();                          // the LDS keys are rewritten next round
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (need[k] == NEED_SPHERE) sphere_pt[k] = mk(map_pm1(w[k][0]), map_pm1(w[k][1]), map_pm1(w[k][2]));
}

template <bool PLDS>
__device__ __forceinline__ void trace_body2(const TraceArgs &a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ CoopLds2 s_coop[MIRT_BLOCK / 64];
    constexpr uint32_t STASH_N = 128u;
    __shared__ StashLds<STASH_N> s_stash[MIRT_BLOCK / 64];
    __shared__ uint32_t s_hist[65];

    if (a.own_chunks == 0u && (uint64_t)blockIdx.x * a.claim >= a.n_paths) return;   // more workgroups than claims (uniform)
    const LdsView lv = lds_view(a.sc, smem);
    if (threadIdx.x < 65) s_hist[threadIdx.x] = 0;
    lds_stage(a.sc, lv);                                            // ends with a barrier
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    auto &stash = s_stash[wave];

    uint32_t cnt = 0, w_next = 0, w_end = 0, own_k = 0;            // wave-uniform bookkeeping, as in trace_body
    auto next_range = [&]() -> bool {
        if (own_k < a.own_chunks) {
            w_next = (own_k * a.grid_waves + (blockIdx.x * (MIRT_BLOCK / 64) + wave)) * 64u; w_end = w_next + 64u;
            ++own_k;
            return true;
        }
        uint32_t b = 0;
        if (lane == 0) b = a.first_claim + atomicAdd(a.cursor, a.claim);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        if (b >= a.n_paths) return false;
        w_next = b; w_end = (a.n_paths - b < a.claim) ? a.n_paths : b + a.claim;
        return true;
    };
    bool exhausted = false;
    // the lane's two paths.  A path slot that is not live still goes through the list walk (its tests cost nothing extra in a
    // wave pass); it keeps a harmless finite ray and takes part in nothing else.
    bool live[2] = { false, false };
    f3 o[2] = { mk(0, 0, 0), mk(0, 0, 0) }, d[2] = { mk(1, 1, 1), mk(1, 1, 1) }, thr[2] = { mk(1.0f, 1.0f, 1.0f), mk(1.0f, 1.0f, 1.0f) };
    uint32_t pid[2] = { 0, 0 }, pixel[2] = { 0, 0 }, sample[2] = { 0, 0 }, depth[2] = { 0, 0 };
    bool odd = false;                                               // wave-uniform: parity of the iteration

    for (;;) {
        // ---- top up the stash: 64 camera rays by the whole wave (HomemadeRayTracer.cpp:114-131)
        if (cnt + 64u <= STASH_N && !exhausted) {
            if (w_next == w_end && !next_range()) exhausted = true;
            if (!exhausted) {
                const uint32_t g = (w_end - w_next < 64u) ? w_end - w_next : 64u;
                if (lane < g) {
                    const TraceArgs &ka = late_args();
                    const uint32_t id = w_next + lane;
                    const uint32_t sl = fast_div(id, ka.by_npix), pl = id - sl * ka.npix;
                    f3 go, gd;
                    generate_ray(ka, ka.pix0 + pl, ka.spp0 + sl, go, gd);       // the list walk never tiles its path ids
                    const uint32_t e = cnt + lane;
                    stash.a[e] = make_float4(go.x, go.y, go.z, gd.x);
                    stash.b[e] = make_float4(gd.y, gd.z, __uint_as_float(id), __uint_as_float(sl));
                }
                w_next += g; cnt += g;
                wave_lds_sync();
            }
        }
        // ---- idle path slots take a fresh path from the top of the stash (slots of path 0 first)
        const unsigned long long idle0 = __ballot(!live[0]), idle1 = __ballot(!live[1]);
        if ((idle0 | idle1) != 0ull && cnt != 0u) {
            const uint32_t n_i0 = (uint32_t)__popcll(idle0), n_idle = n_i0 + (uint32_t)__popcll(idle1);
            const uint32_t rank[2] = { rank_in(idle0), n_i0 + rank_in(idle1) };
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!live[k] && rank[k] < cnt) {
                    const uint32_t e = cnt - 1u - rank[k];
                    const float4 qa = stash.a[e], qb = stash.b[e];
                    o[k] = mk(qa.x, qa.y, qa.z); d[k] = mk(qa.w, qb.x, qb.y); pid[k] = __float_as_uint(qb.z);
                    const uint32_t sl = __float_as_uint(qb.w);
                    sample[k] = a.spp0 + sl; pixel[k] = a.pix0 + (pid[k] - sl * a.npix);
                    thr[k] = mk(1.0f, 1.0f, 1.0f); depth[k] = 0; live[k] = true;
                }
            }
            cnt -= (n_idle < cnt) ? n_idle : cnt;
            wave_lds_sync();                                        // the freed entries are rewritten by the next top-up
        }
        if (__ballot(live[0] | live[1]) == 0ull) break;             // pool exhausted, stash empty, every path ended

        // ---- phase 1: extend both paths (HomemadeRayTracer.cpp:160-162), classify each vertex
        float t_max[2] = { 3.402823466e+38f, 3.402823466e+38f };   // FLT_MAX
        int32_t best[2];
        closest_list2<PLDS>(a.sc, lv, o, d, 0.001f, t_max, best);
        bool done[2] = { false, false };
        uint32_t need[2] = { NEED_NONE, NEED_NONE }, stage[2];
        f3 hp[2], hn[2];                       // the hit's position and normal; hp is the path's radiance where `done`
        uint32_t mi[2] = { 0u, 0u }, kind[2] = { 0u, 0u };   // material index and kind (a pointer kept across the draws would be a generic one: flat loads)
        auto mat_of = [&](uint32_t i) -> const MatRec * {
            if constexpr (PLDS) return (const MatRec *)((const unsigned char *)lv.mats + __umul24(i, (uint32_t)sizeof(MatRec)));
            else return (i < a.sc.lds_mats) ? (const MatRec *)(lv.mats + i) : (a.sc.mats + i);
        };
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            stage[k] = 1u + depth[k];
            hp[k] = o[k]; hn[k] = d[k];
            if (live[k]) {
                if (best[k] < 0) {                                                     // :186
                    hp[k] = thr[k] * mk(a.sc.ambient[0], a.sc.ambient[1], a.sc.ambient[2]); done[k] = true;
                } else {
                    const Prims<PLDS> P(a.sc, lv);
                    Hit h;
                    prim_record(P[best[k]], best[k], o[k], d[k], t_max[k], false, h);
                    if (a.flags & MIRT_FLAG_NORMAL_DISPLAY) {                          // :164-167
                        hp[k] = scale(mk(h.nrm.x + 1.0f, h.nrm.y + 1.0f, h.nrm.z + 1.0f), 0.5f); done[k] = true;
                    } else if (!(depth[k] < a.max_depth)) {                            // :173-174
                        hp[k] = mk(0.0f, 0.0f, 0.0f); done[k] = true;
                    } else {
                        hp[k] = h.pos; hn[k] = h.nrm;
                        mi[k] = (uint32_t)h.mat; kind[k] = mat_of(mi[k])->kind;
                        need[k] = scatter_need(kind[k], d[k], h.nrm, true);
                    }
                }
            }
        }
        // ---- phase 2 (whole wave): the draws of Scatter for both paths, keyed by each path's own depth
        f3 sp[2] = { mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 0.0f) }; float un[2] = { 0.0f, 0.0f };
#if MIRT_POOL_DRAWS
        coop_draws2(s_coop[wave], lane, need, pixel, sample, stage, a.k0, a.k1, sp, un);
#else
        coop_draws<true>(*(CoopLds *)&s_coop[wave], lane, need[0], pixel[0], sample[0], stage[0], a.k0, a.k1, sp[0], un[0]);
        coop_draws<true>(*(CoopLds *)&s_coop[wave], lane, need[1], pixel[1], sample[1], stage[1], a.k0, a.k1, sp[1], un[1]);
#endif
        // ---- phase 3: the rest of Scatter.  Lambertian and light per path; metal and dielectric of both paths in one pass
        bool md[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool shade = live[k] && !done[k];
            md[k] = shade && ((kind[k] & 2u) != 0u);                                 // MIRT_MAT_METAL = 2, MIRT_MAT_DIELECTRIC = 3
            if (shade && !md[k]) {
                Scat s;
                const MatRec *Mk = mat_of(mi[k]);
                scatter_finish<false, 1>(*Mk, mk(Mk->albedo[0], Mk->albedo[1], Mk->albedo[2]), d[k], hp[k], hn[k], sp[k], un[k], s);
                if (s.alive) { o[k] = hp[k]; d[k] = s.dir; thr[k] = thr[k] * s.atten; }
                else { hp[k] = s.emits ? thr[k] * s.emitted : scale(thr[k], 0.0f); done[k] = true; }   // an absorbed path: thr * (0,0,0), :173
            }
        }
        // both paths of the lane need the pass: one is shaded, the other meets its vertex again next time -- path 1 in even
        // iterations, path 0 in odd ones, so that neither starves in a scene of metal and glass only
        const bool both = md[0] & md[1];
        odd = !odd;
        if (md[0] | md[1]) {
            const bool s1 = md[1] & (!md[0] | odd);
            const MatRec *SM = mat_of(s1 ? mi[1] : mi[0]);
            const f3 din = s1 ? d[1] : d[0], pos = s1 ? hp[1] : hp[0], nrm = s1 ? hn[1] : hn[0], spt = s1 ? sp[1] : sp[0];
            const float su = s1 ? un[1] : un[0];
            Scat s;
            scatter_finish<false, 2>(*SM, mk(SM->albedo[0], SM->albedo[1], SM->albedo[2]), din, pos, nrm, spt, su, s);
            if (s1) {
                if (s.alive) { o[1] = pos; d[1] = s.dir; thr[1] = thr[1] * s.atten; }
                else { hp[1] = scale(thr[1], 0.0f); done[1] = true; }
            } else {
                if (s.alive) { o[0] = pos; d[0] = s.dir; thr[0] = thr[0] * s.atten; }
                else { hp[0] = scale(thr[0], 0.0f); done[0] = true; }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool rt = both && ((k == 1) != odd);             // the path of the pair that was not shaded
            if (live[k] && done[k]) {
                store_rad(a.rad, pid[k], hp[k]);
                atomicAdd(&s_hist[depth[k] < 63u ? depth[k] : 63u], 1u);
                if (depth[k] > 63u) atomicAdd(&s_hist[64], depth[k] - 63u);
                live[k] = false;
            }
            if (!rt) ++depth[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 65 && s_hist[threadIdx.x]) atomicAdd(a.hist + threadIdx.x, s_hist[threadIdx.x]);
}
template <bool PLDS>
__global__ void __launch_bounds__(MIRT_BLOCK, MIRT_LIST2_WAVES) k_trace2(TraceArgs a) { trace_body2<PLDS>(a); }

// The list walk is VALU-bound and compiled for 7 waves per SIMD (72 VGPRs, no spill; 8 would spill and lose).  The
// BVH walks are bound by the CU's L1 and indifferent to 6..8 waves: 8 (64 VGPRs) for the plain walks, 6 for the
// yielding one, which keeps its walk state in registers through the shading code.  All launch with a small LDS plan.
template <bool PLDS, bool FULL>
__global__ void __launch_bounds__(MIRT_BLOCK, MIRT_LIST_WAVES) k_trace(TraceArgs a) { trace_body<0, PLDS, FULL>(a); }
template <int ACCEL, bool PLDS, bool FULL>
__global__ void __launch_bounds__(MIRT_BLOCK, ACCEL >= 3 ? MIRT_YIELD_WAVES : MIRT_BVH_WAVES) k_trace_bvh(TraceArgs a) { trace_body<ACCEL, PLDS, FULL>(a); }

// ---------------------------------------------------------------------------------- accumulate
// HomemadeRayTracer.cpp:125-133: col += Sample(...) in sample order.  sum is W*H*3 (interleaved).
__global__ void __launch_bounds__(MIRT_BLOCK) k_accumulate(const RadSlot *__restrict__ rad, float *__restrict__ sum,
                                                           uint32_t pix0, uint32_t npix, uint32_t nsamp, uint32_t overwrite,
                                                           uint32_t tiles_x, uint32_t width)
{
    uint32_t pl = blockIdx.x * MIRT_BLOCK + threadIdx.x;
    if (pl >= npix) return;
    float *s = sum + 3 * (size_t)tile_pixel(pix0 + pl, tiles_x, width);
    f3 acc = overwrite ? mk(0.0f, 0.0f, 0.0f) : mk(s[0], s[1], s[2]);
    const f32x3_a4 *p = (const f32x3_a4 *)(rad + pl);              // sample k of this pixel: k * npix slots further on
    if (overwrite) { const f32x3_t r = p[0]; acc = mk(r.x, r.y, r.z); }
    else {
        uint32_t k = 0;
        for (; k + 8u <= nsamp; k += 8u) {                         // eight loads in flight, then the ordered adds; read once: no reuse
            f32x3_t r[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) r[j] = __builtin_nontemporal_load((const f32x3_a4 *)((const RadSlot *)p + (size_t)(k + j) * npix));
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) acc = acc + mk(r[j].x, r[j].y, r[j].z);
        }
        for (; k < nsamp; ++k) { const f32x3_t r = *(const f32x3_a4 *)((const RadSlot *)p + (size_t)k * npix); acc = acc + mk(r.x, r.y, r.z); }
    }
    s[0] = acc.x; s[1] = acc.y; s[2] = acc.z;
}

// ---------------------------------------------------------------------------------- resolve
// HomemadeRayTracer.cpp:134-138 and OutputImage.cpp:76-83.
__device__ __forceinline__ float clamp01(float v)
{
    float m = (0.0f > v) ? 0.0f : v;        // _mm_max_ps(lo, v)
    return (1.0f < m) ? 1.0f : m;           // _mm_min_ps(hi, m)
}
__device__ __forceinline__ uint32_t to_byte(float c)
{
    float s = c * 255.0f;
    if (!(s == s)) return 0u;               // NaN -> cvttss2si gives 0x80000000, & 0xFF = 0
    return ((uint32_t)s) & 0xFFu;
}
__global__ void __launch_bounds__(MIRT_BLOCK) k_resolve(const float *__restrict__ sum, uint32_t npix, uint32_t spp_total,
                                                        uint32_t flags, float *__restrict__ pixels, uint32_t *__restrict__ rgba8)
{
    uint32_t p = blockIdx.x * MIRT_BLOCK + threadIdx.x;
    if (p >= npix) return;
    f3 col = mk(sum[3 * (size_t)p], sum[3 * (size_t)p + 1], sum[3 * (size_t)p + 2]);
    if (!(flags & MIRT_FLAG_ONE_SPP)) col = scale(col, 1.f / (float)spp_total);      // Vec3.h:30
    col = mk(__builtin_sqrtf(col.x), __builtin_sqrtf(col.y), __builtin_sqrtf(col.z));
    if (pixels) { pixels[3 * (size_t)p] = col.x; pixels[3 * (size_t)p + 1] = col.y; pixels[3 * (size_t)p + 2] = col.z; }
    if (rgba8) rgba8[p] = to_byte(clamp01(col.x)) | (to_byte(clamp01(col.y)) << 8) | (to_byte(clamp01(col.z)) << 16) | 0xFF000000u;
}

// ---------------------------------------------------------------------------------- function-level hooks
__device__ __forceinline__ void store_hit(mirt_hit *o, const Hit &h, bool hit)
{
    if (!hit) { mirt_hit z = {}; z.mat = -1; z.prim = -1; *o = z; return; }
    o->t = h.t; o->pos[0] = h.pos.x; o->pos[1] = h.pos.y; o->pos[2] = h.pos.z;
    o->nrm[0] = h.nrm.x; o->nrm[1] = h.nrm.y; o->nrm[2] = h.nrm.z; o->u = h.u; o->v = h.v;
    o->mat = h.mat; o->prim = h.prim;
}

__global__ void k_aabb_batch(const float *bmin, const float *bmax, uint32_t nbox, const float *org, const float *dir,
                             uint32_t nray, float t_min, float t_max, uint8_t *out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nray) return;
    f3 o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]), d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    for (uint32_t b = 0; b < nbox; ++b) {
        float mn[3] = { bmin[3 * b], bmin[3 * b + 1], bmin[3 * b + 2] }, mx[3] = { bmax[3 * b], bmax[3 * b + 1], bmax[3 * b + 2] };
        out[(size_t)i * nbox + b] = aabb_hit(mn, mx, o, d, t_min, t_max) ? 1 : 0;
    }
}

__global__ void k_prim_hit_batch(SceneDev sc, uint32_t prim, const float *org, const float *dir, uint32_t nray,
                                 float t_min, float t_max, mirt_hit *out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nray) return;
    f3 o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]), d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    float t; Hit h;
    bool hit = prim_t<true>(sc.prims[prim], o, d, t_min, t_max, t, sc.ops);
    if (hit) prim_record<true>(sc.prims[prim], (int32_t)prim, o, d, t, true, h, sc.ops);
    store_hit(out + i, h, hit);
}

template <int ACCEL, bool PLDS>
__global__ void __launch_bounds__(MIRT_BLOCK) k_world_hit_batch(SceneDev sc, const float *org, const float *dir, uint32_t nray,
                                                                float t_min, float t_max, mirt_hit *out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsView lv = lds_view(sc, smem);
    lds_stage(sc, lv);
    for (uint32_t i = blockIdx.x * MIRT_BLOCK + threadIdx.x; i < nray; i += gridDim.x * MIRT_BLOCK) {
        f3 o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]), d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        float tm = t_max;
        int32_t best = ACCEL == 2 ? closest_bvh2<PLDS, true>(sc, lv, o, d, t_min, tm, MIRT_LEAF_WANT)
                     : (ACCEL == 1 ? closest_bvh<PLDS, true>(sc, lv, o, d, t_min, tm) : closest_list<PLDS, true>(sc, lv, o, d, t_min, tm));
        Hit h;
        const Prims<PLDS> P(sc, lv);
        if (best >= 0) prim_record<true>(P[best], best, o, d, tm, true, h, sc.ops);
        store_hit(out + i, h, best >= 0);
    }
}

__global__ void k_scatter_batch(SceneDev sc, const mirt_hit *hits, const float *dir_in, uint32_t n, const uint32_t *draws,
                                uint32_t dpr, uint8_t *alive, float *atten, float *oorg, float *odir, float *oemit, uint32_t *consumed)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mirt_hit hh = hits[i];
    StreamDraws g; g.p = draws + (size_t)i * dpr; g.n = dpr; g.pos = 0;
    f3 pos = mk(hh.pos[0], hh.pos[1], hh.pos[2]), nrm = mk(hh.nrm[0], hh.nrm[1], hh.nrm[2]);
    f3 din = mk(dir_in[3 * i], dir_in[3 * i + 1], dir_in[3 * i + 2]);
    Scat s; s.alive = false;
    const MatRec &mr = sc.mats[hh.mat];
    scatter(mr, albedo_of(mr, sc.texs, sc.texels, hh.u, hh.v), din, pos, nrm, g, s);
    alive[i] = s.alive ? 1 : 0;
    atten[3 * i] = s.atten.x; atten[3 * i + 1] = s.atten.y; atten[3 * i + 2] = s.atten.z;
    oorg[3 * i] = pos.x; oorg[3 * i + 1] = pos.y; oorg[3 * i + 2] = pos.z;
    odir[3 * i] = s.dir.x; odir[3 * i + 1] = s.dir.y; odir[3 * i + 2] = s.dir.z;
    oemit[3 * i] = s.emitted.x; oemit[3 * i + 1] = s.emitted.y; oemit[3 * i + 2] = s.emitted.z;
    if (consumed) consumed[i] = g.pos;
}

__global__ void k_tex_sample(SceneDev sc, uint32_t tex, const float *uv, uint32_t n, float *out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 c = tex_sample(sc.texs[tex], sc.texels, uv[2 * i], uv[2 * i + 1]);
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}

__global__ void k_camera_rays(mirt_camera cam, const float *uv, uint32_t n, const uint32_t *draws, uint32_t dpr,
                              float *oorg, float *odir, uint32_t *consumed)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StreamDraws g; g.p = draws + (size_t)i * dpr; g.n = dpr; g.pos = 0;
    f3 p = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t it = 0; it < MIRT_REJECT_CAP; ++it) {                // Randomizer.h:50-58
        float x = map_pm1(g.next()), y = map_pm1(g.next());
        p = mk(x, y, 0.0f);
        if (!(dot(p, p) >= 1.0f)) break;
    }
    f3 o, d;
    camera_ray(cam, uv[2 * i], uv[2 * i + 1], p, o, d);
    oorg[3 * i] = o.x; oorg[3 * i + 1] = o.y; oorg[3 * i + 2] = o.z;
    odir[3 * i] = d.x; odir[3 * i + 1] = d.y; odir[3 * i + 2] = d.z;
    if (consumed) consumed[i] = g.pos;
}

__global__ void k_philox_batch(const uint32_t *ctr, uint32_t n, uint32_t k0, uint32_t k1, uint32_t *out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4];
    philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, r);
    // the head/tail forms coop_draws uses must give the same block at every level of sharing: a difference inverts the words,
    // which the caller's comparison with the known answers then reports
    uint32_t s1[4], s2[4], s3[4], bad = 0u;
    philox_tail<1>(philox_head<1>(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], k0, k1), ctr[4 * i + 3], k0, k1, s1);
    philox_tail<2>(philox_head<2>(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], k0, k1), ctr[4 * i + 3], k0, k1, s2);
    philox_tail<3>(philox_head<3>(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], k0, k1), ctr[4 * i + 3], k0, k1, s3);
    for (int k = 0; k < 4; ++k) bad |= (s1[k] ^ r[k]) | (s2[k] ^ r[k]) | (s3[k] ^ r[k]);
    const uint32_t flip = bad ? 0xFFFFFFFFu : 0u;
    out[4 * i] = r[0] ^ flip; out[4 * i + 1] = r[1] ^ flip; out[4 * i + 2] = r[2] ^ flip; out[4 * i + 3] = r[3] ^ flip;
}

// arithmetic self-test: op 0 a/b, 1 sqrt(a), 2 pow5f(a), 3 map_unorm(bits a), 4 map_pm1(bits a),
// 5 div_by(div_prepare(b), a), 6 whether (a, b) lies where 5 promises the bits of a / b, 7 sqrt_wave(a), 8 sqrt_in_range(a),
// 9 fast_div(bits a, fastdiv_make(bits b)) as bits
__global__ void k_arith(uint32_t op, const float *x, const float *y, uint32_t n, float *out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y ? y[i] : 0.0f, r;
    switch (op) {
    case 0: r = a / b; break;
    case 1: r = __builtin_sqrtf(a); break;
    case 2: r = pow5f(a); break;
    case 3: r = map_unorm(__float_as_uint(a)); break;
    case 4: r = map_pm1(__float_as_uint(a)); break;
    case 5: r = div_by(div_prepare(b), a); break;                        // the shared-denominator division of rect_loop, raw
    case 6: r = (div_den_ok(b) & div_num_ok(a)) ? 1.0f : 0.0f; break;       // inside the operand range where 5 promises the bits of a / b
    case 7: r = sqrt_wave(a); break;                                      // the render kernels' square root (lean form per wave, rt_device.h)
    case 9: r = __uint_as_float(fast_div(__float_as_uint(a), fastdiv_make(__float_as_uint(b)))); break;   // path id / (pixels of a batch, image width)
    default: r = sqrt_in_range(a); break;                                 // 8: the lean form alone, promised for [2^-60, 2^60]
    }
    out[i] = r;
}

} // namespace mirt
