// scene_host.h -- host-side scene preparation of libmirt: validation of a mirt_scene, object
// bounding boxes, the library's own BVH builder (used when the caller passes no tree), breadth-
// first renumbering and packing into the device records of rt_device.h.
//
// Bounding boxes restate IHitable::BoundingBox (Hitables.cpp:56-60 sphere, :117-128 rect,
// :152-161 combo, :175-180 translate, :208-238 rotate) and CombineAABB (AABB.h:19-34).
// File:line citations are relative to /root/reference/RayTracer/RayTracer/.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/mirt.h"

namespace mirt { namespace host {

struct Box { float mn[3], mx[3]; };

inline Box combine(const Box &a, const Box &b)                        // AABB.h:19-34
{
    Box r;
    for (int i = 0; i < 3; ++i) { r.mn[i] = (a.mn[i] < b.mn[i]) ? a.mn[i] : b.mn[i]; r.mx[i] = (a.mx[i] > b.mx[i]) ? a.mx[i] : b.mx[i]; }
    return r;
}
inline Box rect_box(const mirt_scene &s, uint32_t p)                  // Hitables.cpp:117-128
{
    uint32_t ax = s.prim_axes[p]; const float *g = s.prim_geom + 5 * (size_t)p;
    uint32_t a = ax & 3, b = (ax >> 2) & 3, c = (ax >> 4) & 3;
    Box r = {};
    r.mn[a] = g[0]; r.mx[a] = g[1]; r.mn[b] = g[2]; r.mx[b] = g[3];
    r.mn[c] = g[4] - 0.0001f; r.mx[c] = g[4] + 0.0001f;
    return r;
}
inline Box rot_box(const Box &in, uint32_t axis, float sn, float cs)  // Hitables.cpp:188-238
{
    uint32_t aA, bA; float OP;
    if (axis == 0) { aA = 1; bA = 2; OP = -1.0f; } else if (axis == 1) { aA = 0; bA = 2; OP = 1.0f; } else { aA = 0; bA = 1; OP = -1.0f; }
    Box r; for (int i = 0; i < 3; ++i) { r.mn[i] = FLT_MAX; r.mx[i] = -FLT_MAX; }
    for (uint32_t i = 0; i < 2; ++i) for (uint32_t j = 0; j < 2; ++j) for (uint32_t k = 0; k < 2; ++k) {
        float a = i * in.mx[aA] + (1 - i) * in.mn[aA];
        float b = j * in.mx[bA] + (1 - j) * in.mn[bA];
        float c = k * in.mx[axis] + (1 - k) * in.mn[axis];
        float t[3]; t[aA] = cs * a + OP * sn * b; t[bA] = -OP * sn * a + cs * b; t[axis] = c;
        for (int x = 0; x < 3; ++x) { if (t[x] > r.mx[x]) r.mx[x] = t[x]; if (t[x] < r.mn[x]) r.mn[x] = t[x]; }
    }
    return r;
}
inline bool chain_form(const mirt_scene &s, uint32_t p) { return s.prim_chain && s.prim_chain[p] != 0; }
// A chain-form primitive's box: the leaf's, carried through its instances from the innermost out (Hitables.cpp:175-180, :208-238).
// A sphere's is the box around ITS CENTRE: the reference's SphereHitable::BoundingBox (Hitables.cpp:56-60) forgets m_center, which
// is harmless for its factories (centre 0) and wrong for any other sphere; these boxes serve the library's own builder and the
// object boxes of the near-first walk, which must contain what they guard.
inline Box prim_box_chain(const mirt_scene &s, uint32_t p)
{
    Box b;
    if (s.prim_kind[p] == MIRT_PRIM_SPHERE) {
        const float r = s.prim_geom[5 * (size_t)p];
        for (int i = 0; i < 3; ++i) { b.mn[i] = s.prim_center[3 * (size_t)p + i] - r; b.mx[i] = s.prim_center[3 * (size_t)p + i] + r; }
    } else b = rect_box(s, p);
    const uint32_t first = s.prim_chain[p] - 1u, n = s.prim_nops[p];
    for (uint32_t k = n; k-- > 0;) {
        const uint32_t kind = s.op_kind[first + k]; const float *a = s.op_arg + 3 * (size_t)(first + k);
        if (kind == MIRT_OP_TRANSLATE) { for (int i = 0; i < 3; ++i) { b.mn[i] = b.mn[i] + a[i]; b.mx[i] = b.mx[i] + a[i]; } }
        else b = rot_box(b, kind - 1u, a[0], a[1]);
    }
    return b;
}
inline Box object_box_factory(const mirt_scene &s, uint32_t first, uint32_t cnt)
{
    Box b;
    if (s.prim_kind[first] == MIRT_PRIM_SPHERE) {                      // Hitables.cpp:56-60
        float r = s.prim_geom[5 * (size_t)first];
        for (int i = 0; i < 3; ++i) { b.mn[i] = 0.0f - r; b.mx[i] = r; }
    } else {
        b = rect_box(s, first);
        for (uint32_t i = 1; i < cnt; ++i) b = combine(rect_box(s, first + i), b);   // Hitables.cpp:152-161
        if ((s.prim_axes[first] >> 7) & 1) {
            const float *q = s.prim_rot + 6 * (size_t)first;
            b = rot_box(b, 0, q[0], q[1]); b = rot_box(b, 1, q[2], q[3]); b = rot_box(b, 2, q[4], q[5]);
        }
    }
    for (int i = 0; i < 3; ++i) { b.mn[i] = b.mn[i] + s.prim_off[3 * (size_t)first + i]; b.mx[i] = b.mx[i] + s.prim_off[3 * (size_t)first + i]; }  // :175-180
    return b;
}
// An object of factory-form primitives is one of the factories' shapes (a sphere, a rect, a cube's six faces inside ONE chain): its
// box is built the way the reference builds it, combo first, then the chain.  An object with a chain-form primitive is a general
// combo: the union, in order, of its primitives' boxes (Hitables.cpp:152-161).
inline Box object_box(const mirt_scene &s, uint32_t obj)
{
    const uint32_t first = s.obj_first[obj], cnt = s.obj_count[obj];
    bool any_chain = false;
    for (uint32_t i = 0; i < cnt; ++i) any_chain |= chain_form(s, first + i);
    if (!any_chain) return object_box_factory(s, first, cnt);
    Box b = chain_form(s, first) ? prim_box_chain(s, first) : object_box_factory(s, first, 1);
    for (uint32_t i = 1; i < cnt; ++i) b = combine(chain_form(s, first + i) ? prim_box_chain(s, first + i) : object_box_factory(s, first + i, 1), b);
    return b;
}

struct Tree {
    std::vector<Box> box; std::vector<int32_t> left, right; int32_t root = MIRT_CHILD_NONE;
    uint32_t depth = 0;
    std::vector<Box> obox;          // object boxes (children of leaves), filled by prepare_tree
};

// The library's own builder, O(N log N): binned surface-area heuristic over the object boxes -- the cheapest cut between bins
// of the centroid bounds on any of the three axes; 16 to 128 bins, an eighth of the node's objects; median split on the widest
// axis when the bins do not separate anything or the tree gets deeper than `kSahDepth`, which bounds the traversal stack.
// Measured against round 1's builder (16 bins, widest axis only, SAH down to level 16): 485 spheres 1200x800@128 17.40 -> 16.51 ms,
// ~1 M spheres 1920x1080@32 18.6 -> 16.3 ms; an exact sweep over every cut for nodes of up to 4096 objects changed nothing more,
// and tree rotations afterwards (Kensler 2008) shrank the summed inner area by 0.1 % and 0.01 %: neither is kept.
// Leaves are shaped like the reference's (a node whose children are objects, SimpleObject.cpp:308-322).
// Any valid BVH yields the same closest t; only the winner of an exact tie depends on the order.
struct Builder {
    static constexpr int kBins = 128, kSahDepth = 32;
    const mirt_scene &s; const std::vector<Box> &ob; Tree &t;
    static float area(const Box &b)
    {
        float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
    float centroid(uint32_t o, int a) const { return 0.5f * (ob[o].mn[a] + ob[o].mx[a]); }
    int32_t build(uint32_t *o, size_t n, int depth = 0)
    {
        int32_t me = (int32_t)t.box.size();
        t.box.push_back(Box{}); t.left.push_back(MIRT_CHILD_NONE); t.right.push_back(MIRT_CHILD_NONE);
        int32_t l, r;
        if (n == 1) { l = ~(int32_t)o[0]; r = MIRT_CHILD_NONE; }
        else if (n == 2) { l = ~(int32_t)o[0]; r = ~(int32_t)o[1]; }
        else {
            float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            for (size_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { float c = centroid(o[i], a); lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
            int ax = 0; if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1; if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;   // widest: the median's axis
            size_t half = 0;
            if (depth < kSahDepth) {
                // the cheapest of the nb - 1 cuts on each of the three axes (cost = area x count of either side)
                float best = FLT_MAX; int cut = -1, cut_ax = -1;
                const int nb = (int)std::min<size_t>(kBins, std::max<size_t>(16, n / 8));      // bins: an eighth of the objects, 16 to 128
                for (int x = 0; x < 3; ++x) {
                    const float ext = hi[x] - lo[x];
                    if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
                    Box bb[kBins]; size_t cnt[kBins] = {};
                    auto bin_of = [&](uint32_t q) { int b = (int)((centroid(q, x) - lo[x]) / ext * nb); return b < 0 ? 0 : (b >= nb ? nb - 1 : b); };
                    for (size_t i = 0; i < n; ++i) { int b = bin_of(o[i]); bb[b] = cnt[b] ? combine(bb[b], ob[o[i]]) : ob[o[i]]; cnt[b]++; }
                    float rarea[kBins]; size_t rcnt[kBins]; Box acc{}; size_t c = 0;
                    for (int b = nb - 1; b > 0; --b) { if (cnt[b]) { acc = c ? combine(acc, bb[b]) : bb[b]; c += cnt[b]; } rarea[b] = c ? area(acc) : 0.0f; rcnt[b] = c; }
                    c = 0;
                    for (int b = 0; b < nb - 1; ++b) {
                        if (cnt[b]) { acc = c ? combine(acc, bb[b]) : bb[b]; c += cnt[b]; }
                        if (!c || !rcnt[b + 1]) continue;
                        float cost = area(acc) * (float)c + rarea[b + 1] * (float)rcnt[b + 1];
                        if (cost < best) { best = cost; cut = b; cut_ax = x; }
                    }
                }
                if (cut >= 0) {
                    const float ext = hi[cut_ax] - lo[cut_ax];
                    auto bin_of = [&](uint32_t q) { int b = (int)((centroid(q, cut_ax) - lo[cut_ax]) / ext * nb); return b < 0 ? 0 : (b >= nb ? nb - 1 : b); };
                    half = (size_t)(std::partition(o, o + n, [&](uint32_t q) { return bin_of(q) <= cut; }) - o);
                }
            }
            if (half == 0 || half == n) {                       // median on the same axis
                half = n / 2;
                std::nth_element(o, o + half, o + n, [&](uint32_t x, uint32_t y) { float cx = centroid(x, ax), cy = centroid(y, ax); return cx < cy || (cx == cy && x < y); });
            }
            // a side of one object becomes a child directly (a node over a single object would only add a box test)
            l = half == 1 ? ~(int32_t)o[0] : build(o, half, depth + 1);
            r = n - half == 1 ? ~(int32_t)o[half] : build(o + half, n - half, depth + 1);
        }
        auto cb = [&](int32_t c) { return c < 0 ? ob[~c] : t.box[c]; };
        t.box[me] = (r == MIRT_CHILD_NONE) ? cb(l) : combine(cb(l), cb(r));
        t.left[me] = l; t.right[me] = r;
        return me;
    }
};

inline std::string validate(const mirt_scene &s)
{
    if (!s.n_prims || !s.n_objects || !s.n_mats) return "scene needs at least one primitive, object and material";
    if (!s.prim_kind || !s.prim_off || !s.prim_geom || !s.prim_axes || !s.prim_rot || !s.prim_mat || !s.obj_first || !s.obj_count ||
        !s.mat_kind || !s.mat_albedo || !s.mat_fuzz || !s.mat_ior || !s.mat_emit) return "null scene array";
    uint32_t next = 0;
    for (uint32_t o = 0; o < s.n_objects; ++o) {
        if (s.obj_first[o] != next || s.obj_count[o] == 0) return "objects must be consecutive non-empty runs of primitives";
        if (s.obj_count[o] > s.n_prims - next) return "objects do not cover the primitives";
        next += s.obj_count[o];
        // several factory-form primitives in one object are a cube: six rects inside ONE chain (SimpleObject.cpp:261-276), which is
        // how their box is built; a general combo brings its primitives in chain form
        bool chains = false;
        for (uint32_t i = 0; i < s.obj_count[o]; ++i) chains |= s.prim_chain && s.prim_chain[s.obj_first[o] + i] != 0;
        for (uint32_t i = 1; !chains && i < s.obj_count[o]; ++i) {
            const uint32_t f = s.obj_first[o], p = f + i;
            if (s.prim_kind[p] != MIRT_PRIM_RECT || s.prim_kind[f] != MIRT_PRIM_RECT) return "a multi-primitive object of factory-form primitives must be rects (HitableCombo of a cube); general combos use chain form";
            bool same = ((s.prim_axes[p] >> 7) & 1) == ((s.prim_axes[f] >> 7) & 1);
            for (int k = 0; k < 3; ++k) same &= s.prim_off[3 * (size_t)p + k] == s.prim_off[3 * (size_t)f + k];
            for (int k = 0; k < 6; ++k) same &= s.prim_rot[6 * (size_t)p + k] == s.prim_rot[6 * (size_t)f + k];
            if (!same) return "the factory-form primitives of one object must share their instance chain (a cube's faces do); general combos use chain form";
        }
    }
    if (next != s.n_prims) return "objects do not cover the primitives";
    for (uint32_t p = 0; p < s.n_prims; ++p) {
        if (s.prim_kind[p] > 1) return "bad primitive kind";
        if (s.prim_mat[p] >= s.n_mats) return "material index out of range";
        if (s.prim_kind[p] == MIRT_PRIM_RECT) {
            uint32_t ax = s.prim_axes[p], a = ax & 3, b = (ax >> 2) & 3, c = (ax >> 4) & 3;
            if (a > 2 || b > 2 || c > 2 || a == b || a == c || b == c) return "bad rect axes";
        }
    }
    if (s.prim_chain) {
        if (!s.prim_nops || !s.prim_center || (s.n_ops && (!s.op_kind || !s.op_arg))) return "null instance-chain array";
        for (uint32_t p = 0; p < s.n_prims; ++p) {
            if (!s.prim_chain[p]) continue;
            if (s.prim_nops[p] > MIRT_MAX_CHAIN_OPS) return "instance chain longer than MIRT_MAX_CHAIN_OPS";
            if ((uint64_t)(s.prim_chain[p] - 1u) + s.prim_nops[p] > s.n_ops) return "instance chain out of range";
            for (uint32_t k = 0; k < s.prim_nops[p]; ++k) if (s.op_kind[s.prim_chain[p] - 1u + k] > MIRT_OP_ROTATE_Z) return "bad instance op";
        }
    }
    for (uint32_t m = 0; m < s.n_mats; ++m) if (s.mat_kind[m] > 3) return "bad material kind";
    if (s.n_textures) {
        if (!s.tex_width || !s.tex_height || !s.tex_offset || !s.tex_rgba8 || !s.mat_texture) return "null texture array";
        for (uint32_t t = 0; t < s.n_textures; ++t) {
            if (!s.tex_width[t] || !s.tex_height[t]) return "empty texture";
            if ((uint64_t)s.tex_offset[t] + (uint64_t)s.tex_width[t] * s.tex_height[t] > 0xFFFFFFFFull) return "textures larger than 2^32 texels";
        }
        for (uint32_t m = 0; m < s.n_mats; ++m) if (s.mat_texture[m] >= (int32_t)s.n_textures) return "texture index out of range";
    }
    if (s.n_nodes) {
        if (!s.node_min || !s.node_max || !s.node_left || !s.node_right) return "null BVH array";
        if (s.root < 0 || (uint32_t)s.root >= s.n_nodes) return "BVH root out of range";
    }
    return "";
}

// Take the caller's tree (or build one), check it is a tree over valid indices, renumber the nodes
// breadth-first (so "index < K" = the top levels, staged in LDS) and measure its depth.
inline std::string prepare_tree(const mirt_scene &s, Tree &out)
{
    Tree t;
    out.obox.resize(s.n_objects);
    for (uint32_t o = 0; o < s.n_objects; ++o) out.obox[o] = object_box(s, o);
    if (s.n_nodes) {
        t.box.resize(s.n_nodes); t.left.assign(s.node_left, s.node_left + s.n_nodes); t.right.assign(s.node_right, s.node_right + s.n_nodes);
        for (uint32_t i = 0; i < s.n_nodes; ++i) for (int a = 0; a < 3; ++a) { t.box[i].mn[a] = s.node_min[3 * (size_t)i + a]; t.box[i].mx[a] = s.node_max[3 * (size_t)i + a]; }
        t.root = s.root;
    } else {
        const std::vector<Box> &ob = out.obox;
        std::vector<uint32_t> idx(s.n_objects);
        for (uint32_t o = 0; o < s.n_objects; ++o) idx[o] = o;
        Builder b{ s, ob, t };
        t.root = b.build(idx.data(), idx.size());
    }
    // BFS renumber + validation
    size_t n = t.box.size();
    std::vector<int32_t> order; order.reserve(n);
    std::vector<int32_t> newid(n, -1);
    std::vector<uint32_t> level;
    order.push_back(t.root); newid[t.root] = 0; level.push_back(1);
    uint32_t depth = 1;
    for (size_t h = 0; h < order.size(); ++h) {
        int32_t c = order[h];
        int32_t ch[2] = { t.left[c], t.right[c] };
        for (int k = 0; k < 2; ++k) {
            int32_t x = ch[k];
            if (x == MIRT_CHILD_NONE) continue;
            if (x < 0) { if ((uint32_t)~x >= s.n_objects) return "BVH leaf object out of range"; continue; }
            if ((size_t)x >= n) return "BVH child out of range";
            if (newid[x] >= 0) return "BVH is not a tree (node reached twice)";
            newid[x] = (int32_t)order.size(); order.push_back(x); level.push_back(level[h] + 1);
            depth = std::max(depth, level[h] + 1);
        }
    }
    out.box.resize(order.size()); out.left.resize(order.size()); out.right.resize(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        int32_t c = order[i];
        out.box[i] = t.box[c];
        out.left[i] = t.left[c] >= 0 ? newid[t.left[c]] : t.left[c];
        out.right[i] = t.right[c] >= 0 ? newid[t.right[c]] : t.right[c];
    }
    out.root = 0; out.depth = depth;
    return "";
}

}} // namespace mirt::host
