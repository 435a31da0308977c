#include <cstdio>
#include <cstring>
#include <random>
#include "scene_host.h"
using namespace mirt::host;
int main()
{
    std::mt19937 rng(99);
    auto U = [&](uint32_t n) { return n ? rng() % n : 0u; };
    size_t ok = 0, bad = 0;
    for (int it = 0; it < 30000; ++it) {
        uint32_t no = 1 + U(12), np = 0;
        std::vector<uint32_t> of, oc, kind, axes, mat;
        for (uint32_t o = 0; o < no; ++o) { uint32_t c = (U(5) == 0) ? 6 : 1; of.push_back(np); oc.push_back(c); bool sph = c == 1 && U(2); for (uint32_t i = 0; i < c; ++i) { kind.push_back(sph ? 0 : 1); uint32_t a = U(3), b = (a + 1 + U(2)) % 3, cc = 3 - a - b; axes.push_back(a | b << 2 | cc << 4 | U(2) << 6 | U(2) << 7); mat.push_back(U(3)); } np += c; }
        std::vector<float> off(3 * np), geom(5 * np), rot(6 * np);
        for (auto &x : off) x = (float)(int)U(20) - 10.0f;
        for (uint32_t o = 0; o < no; ++o) for (uint32_t i = 1; i < oc[o]; ++i) { for (int k = 0; k < 3; ++k) off[3 * (of[o] + i) + k] = off[3 * of[o] + k]; axes[of[o] + i] = (axes[of[o] + i] & ~128u) | (axes[of[o]] & 128u); }   // a cube's faces share their chain
        for (uint32_t p = 0; p < np; ++p) { geom[5 * p] = kind[p] ? -1.0f : 0.5f + U(3); geom[5 * p + 1] = 1.0f; geom[5 * p + 2] = -1.0f; geom[5 * p + 3] = 1.0f; geom[5 * p + 4] = 0.0f; for (int k = 0; k < 3; ++k) { rot[6 * p + 2 * k] = 0.0f; rot[6 * p + 2 * k + 1] = 1.0f; } }
        uint32_t mk[3] = { 1, 2, 3 }; float alb[9] = {}, fz[3] = {}, ior[3] = {}, em[9] = {};
        mirt_scene s{};
        s.n_prims = np; s.prim_kind = kind.data(); s.prim_off = off.data(); s.prim_geom = geom.data(); s.prim_axes = axes.data(); s.prim_rot = rot.data(); s.prim_mat = mat.data();
        s.n_objects = no; s.obj_first = of.data(); s.obj_count = oc.data(); s.n_mats = 3; s.mat_kind = mk; s.mat_albedo = alb; s.mat_fuzz = fz; s.mat_ior = ior; s.mat_emit = em;
        // chain-form primitives a third of the time: mostly valid op ranges, now and then out of range, too long, or a bad op kind
        std::vector<uint32_t> pchain(np, 0), pnops(np, 0), okind; std::vector<float> pcenter(3 * np, 0.25f), oarg;
        if (U(3) == 0) {
            for (uint32_t p = 0; p < np; ++p) if (U(2)) { uint32_t n = U(9); pchain[p] = (uint32_t)okind.size() + 1; pnops[p] = n; for (uint32_t k = 0; k < n; ++k) { okind.push_back(U(4)); oarg.push_back(0.5f); oarg.push_back(0.8f); oarg.push_back(0.0f); } }
            if (okind.empty()) { okind.push_back(0); oarg.resize(3, 0.0f); }
            s.prim_chain = pchain.data(); s.prim_nops = pnops.data(); s.prim_center = pcenter.data(); s.n_ops = (uint32_t)(oarg.size() / 3); s.op_kind = okind.data(); s.op_arg = oarg.data();
            if (U(6) == 0 && np) { uint32_t p = U(np); pchain[p] = 1 + U(s.n_ops + 3); pnops[p] = U(12); }
            if (U(8) == 0) okind[U((uint32_t)okind.size())] = 4 + U(100);
            if (U(8) == 0) s.n_ops = U(s.n_ops + 1);
        }
        // a tree: valid (own builder) half of the time, then corrupted
        std::vector<float> nmin, nmax; std::vector<int32_t> l, r;
        Tree t0;
        if (U(2)) {
            s.n_nodes = 0;
            if (!validate(s).empty()) { ++bad; continue; }          // the chain arrays may already be broken
            if (!prepare_tree(s, t0).empty()) { printf("own tree rejected\n"); return 2; }
            size_t n = t0.box.size();
            nmin.resize(3 * n); nmax.resize(3 * n); l = t0.left; r = t0.right;
            for (size_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { nmin[3 * i + a] = t0.box[i].mn[a]; nmax[3 * i + a] = t0.box[i].mx[a]; }
            if (n) {
                int muts = U(3);
                for (int m = 0; m < muts; ++m) { auto &v = U(2) ? l : r; v[U((uint32_t)n)] = (int32_t)(U(4) == 0 ? rng() : (U(2) ? (int32_t)U((uint32_t)n + 2) : ~(int32_t)U(no + 2))); }
                s.n_nodes = (uint32_t)n; s.node_min = nmin.data(); s.node_max = nmax.data(); s.node_left = l.data(); s.node_right = r.data(); s.root = U(8) == 0 ? (int32_t)rng() : t0.root;
            }
        }
        if (U(10) == 0 && np) mat[U(np)] = 7;
        if (U(10) == 0) oc[U(no)] += 1;
        Tree t;
        std::string why = validate(s);
        if (why.empty()) why = prepare_tree(s, t);
        if (why.empty()) {
            ++ok;
            if (t.depth > 64 + no) { printf("depth %u\n", t.depth); return 3; }
            for (size_t i = 0; i < t.left.size(); ++i) for (int32_t c : { t.left[i], t.right[i] }) if (c != MIRT_CHILD_NONE && !(c >= 0 ? (size_t)c < t.left.size() : (uint32_t)~c < no)) { printf("bad child survived\n"); return 4; }
        } else ++bad;
    }
    printf("accepted %zu, rejected %zu\n", ok, bad);
    return 0;
}
