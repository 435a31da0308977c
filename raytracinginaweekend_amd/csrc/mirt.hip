// mirt.hip -- C ABI of libmirt.so (include/mirt.h): context, scene upload, the render
// loop that drives the gfx950 kernels of rt_kernels.h, and the function-level parity entry points.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "rt_kernels.h"
#include "scene_host.h"

using namespace mirt;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t n)
    {
        if (n <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; return -1; }
        bytes = n; return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// mirt_render_device_noalloc: the persistent engine's scratch (radiance slots, claim cursors + path-length histograms,
// the HBM part of the traversal stacks) in a file-scope device array instead of hipMalloc'ed buffers.  Every GPU the
// library runs on holds its own copy until the process ends.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress: the address kernels and hipMemsetAsync get is g_scratch_ptr[device], looked up once per
// device in mirt_create with that device current.
constexpr size_t SCRATCH_RAD_PATHS = 2u << 20;                       // 2 Mi paths per batch: 24 MiB of 12-byte slots
constexpr size_t SCRATCH_RAD_BYTES = SCRATCH_RAD_PATHS * sizeof(RadSlot);
constexpr size_t SCRATCH_COUNTS_BYTES = 1u << 20;                    // 66 words per batch: up to 3971 batches per call
constexpr size_t SCRATCH_SPILL_BYTES = 31u << 20;                    // traversal-stack levels beyond the LDS part
constexpr size_t SCRATCH_BYTES = SCRATCH_RAD_BYTES + SCRATCH_COUNTS_BYTES + SCRATCH_SPILL_BYTES;   // 56 MiB
constexpr int SCRATCH_MAX_DEVICES = 64;
__device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
unsigned char *g_scratch_ptr[SCRATCH_MAX_DEVICES];                   // written under g_scratch_mu, never cleared
std::mutex g_scratch_mu;

// Where one render keeps its scratch: hipMalloc'ed buffers of the context (grown on demand) or a carving of `scratch`.
struct Workspace {
    RadSlot *rad; size_t rad_paths;
    uint32_t *counts; size_t counts_bytes;
    int32_t *spill; size_t spill_bytes;
};

} // namespace

struct mirt_ctx {
    int device = 0;
    std::string err;
    bool have_scene = false, have_camera = false;
    SceneDev sc{};                  // LDS plan for the list / reference-order walk (32-byte nodes)
    SceneDev sc2{};                 // LDS plan for the near-first walk (64-byte nodes)
    uint32_t lds_bytes = 0, lds_bytes2 = 0, lds_stack_bytes = 0;   // the stack comes last in the plan: the list walk launches without it
    bool prims_in_lds = false;      // primitive AND material tables staged in LDS: the PLDS kernel variants
    bool small_scene = false;       // the primitive table is small enough for the list walk (ACCEL_AUTO)
    mirt_camera cam{};
    DevBuf d_prims, d_lprims, d_nodes, d_nodes2, d_objs, d_mats, d_texs, d_texels;
    DevBuf d_mdq, d_rdy;            // MIRT_DEFER_MD builds: per-wave queues of the second (metal/dielectric) stream
    DevBuf d_sph4;                  // SceneDev::sph4
    DevBuf d_ops;                   // SceneDev::ops: the instances of chain-form primitives
    bool has_chains = false;        // the scene holds chain-form primitives: the FULL kernel variants
    DevBuf d_spill;                 // traversal-stack levels beyond SceneDev::stack_lds, [level][thread of the grid]
    DevBuf d_prog_sum;              // mirt_render_progressive: the radiance sum kept between calls
    uint32_t prog_w = 0, prog_h = 0, prog_spp = 0;
    DevBuf d_queues, d_rad, d_counts, d_len, d_tmp_sum, d_tmp_pix, d_tmp_rgba;
    uint64_t max_paths = 1ull << 30;        // 1 Gi paths per batch = 12.9 GB of radiance slots when fully used (a batch costs ~0.3 ms of tail)
    bool time_launches = false;
    uint32_t blocks_per_cu = 0;             // 0: as many workgroups per CU as the occupancy query reports
    uint32_t lds_budget = 4u * 1024u;      // per workgroup; takes effect at the next mirt_upload_scene
    uint32_t paths_per_lane = 1;            // list walk: 1 = k_trace; 2 = k_trace2 (two paths in every lane share the passes few lanes need: measured 6 % slower, kept under test)
    uint32_t engine = 1;                    // 1: persistent k_trace (no path queues), 0: wavefront k_bounce per depth
    uint32_t lds_prims = 4096u;             // primitives are staged in LDS if they take no more than this; likewise
    int leaf_wait = -1;                    // near-first walks: lanes waiting on a leaf that start an object-test pass (rt_kernels.h closest_bvh2); -1: by the tree's size
    int32_t lds_top = -1;                   // sliced near-first walk: top-of-tree nodes read from LDS (0..256; -1: 32 for trees of 65536 nodes and more, else 0);
                                            // takes effect at the next mirt_upload_scene
    int32_t walk_yield = -1;                // near-first walks in slices that end when this many lanes of a wave have finished;
                                            // 0: off, -1 (default): 56 for batches of 2 Mi paths or more
    uint32_t tile_paths = 1;                // persistent engine, BVH walks: path ids over 8x8 pixel tiles (option tile_paths)
    uint32_t lds_mats = 1024u;              // the material table is staged in LDS if it is no larger than this; likewise
    uint32_t stack_lds = 5;                // traversal-stack levels kept in LDS (1 KiB each; round 3's last sweep, 2/3/4/5/6/8 levels: 485 spheres 12.44/12.21/12.16/12.14/12.10/12.58 ms,
                                           // ~1 M spheres 13.32/13.04/12.95/12.78/13.10/16.3 ms: profiles/r03_walk_thresholds.txt); likewise
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 256;
    unsigned char *scratch_dev = nullptr;   // this device's copy of `scratch` (hipGetSymbolAddress in mirt_create)
};

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return MIRT_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

static int fail(mirt_ctx *c, int code, const std::string &msg) { c->err = msg; return code; }

extern "C" int mirt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int mirt_runtime_version(int *hip_runtime, int *hip_driver)
{
    int rt = 0, drv = 0;
    if (hipRuntimeGetVersion(&rt) != hipSuccess) { g_create_error = "hipRuntimeGetVersion failed"; return MIRT_ERR_HIP; }
    if (hipDriverGetVersion(&drv) != hipSuccess) drv = 0;      // no driver in a container without a GPU: not an error here
    if (hip_runtime) *hip_runtime = rt;
    if (hip_driver) *hip_driver = drv;
    return MIRT_OK;
}

extern "C" uint32_t mirt_abi_version(void) { return MIRT_ABI_VERSION; }

extern "C" int mirt_check_abi(uint32_t version, size_t sizeof_scene, size_t sizeof_camera, size_t sizeof_hit, size_t sizeof_render_params, size_t sizeof_stats)
{
    const bool ok = version == MIRT_ABI_VERSION && sizeof_scene == sizeof(mirt_scene) && sizeof_camera == sizeof(mirt_camera) && sizeof_hit == sizeof(mirt_hit) &&
                    sizeof_render_params == sizeof(mirt_render_params) && sizeof_stats == sizeof(mirt_stats);
    if (!ok) {
        char buf[256];
        snprintf(buf, sizeof buf, "mirt_check_abi: caller built against ABI %u (scene %zu, camera %zu, hit %zu, params %zu, stats %zu bytes), library is ABI %u (%zu, %zu, %zu, %zu, %zu)",
                 version, sizeof_scene, sizeof_camera, sizeof_hit, sizeof_render_params, sizeof_stats, (unsigned)MIRT_ABI_VERSION, sizeof(mirt_scene), sizeof(mirt_camera),
                 sizeof(mirt_hit), sizeof(mirt_render_params), sizeof(mirt_stats));
        g_create_error = buf;                           // mirt_last_error(NULL)
    }
    return ok ? MIRT_OK : MIRT_ERR_ARG;
}

// struct_size of the in/out structs (include/mirt.h, "ABI guard"): a host compiled against another revision of the
// header gets MIRT_ERR_ARG before anything is read or written through the pointer.
static int size_mismatch(mirt_ctx *c, const char *what, uint32_t got, size_t want)
{
    const std::string msg = std::string(what) + ".struct_size is " + std::to_string(got) + ", this library was built with " + std::to_string(want) +
                            " (set it to sizeof, and rebuild the host against this include/mirt.h)";
    if (c) c->err = msg; else g_create_error = msg;
    return MIRT_ERR_ARG;
}
#define CHECK_SIZE(ctx, ptr, type) do { if ((ptr) && (ptr)->struct_size != sizeof(type)) return size_mismatch(ctx, #type, (ptr)->struct_size, sizeof(type)); } while (0)

extern "C" int mirt_create(int device_id, mirt_ctx **out)
{
    if (!out) { g_create_error = "mirt_create: out is NULL"; return MIRT_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = "mirt_create: no HIP device (the path tracer has no CPU fallback)"; return MIRT_ERR_HIP; }
    if (device_id < 0 || device_id >= n) { g_create_error = "mirt_create: device id out of range"; return MIRT_ERR_ARG; }
    if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return MIRT_ERR_HIP; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) { g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return MIRT_ERR_HIP; }
    mirt_ctx *c = new mirt_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { g_create_error = "hipEventCreate failed"; delete c; return MIRT_ERR_HIP; }
    if (device_id < SCRATCH_MAX_DEVICES) {                      // device `device_id` is current (hipSetDevice above)
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[device_id]) {
            void *sp = nullptr;
            if ((e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch))) != hipSuccess || !sp) {
                g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
                (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1); delete c; return MIRT_ERR_HIP;
            }
            g_scratch_ptr[device_id] = (unsigned char *)sp;
        }
        c->scratch_dev = g_scratch_ptr[device_id];
    }
    *out = c;
    return MIRT_OK;
}

extern "C" void mirt_destroy(mirt_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    DevBuf *bufs[] = { &c->d_prims, &c->d_lprims, &c->d_nodes, &c->d_nodes2, &c->d_objs, &c->d_mats, &c->d_texs, &c->d_texels, &c->d_sph4, &c->d_ops, &c->d_spill, &c->d_mdq, &c->d_rdy, &c->d_prog_sum, &c->d_queues, &c->d_rad, &c->d_counts, &c->d_len, &c->d_tmp_sum, &c->d_tmp_pix, &c->d_tmp_rgba };
    for (DevBuf *b : bufs) b->release();
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

extern "C" const char *mirt_last_error(const mirt_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

extern "C" int mirt_set_option(mirt_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return MIRT_ERR_ARG;
    if (!strcmp(name, "max_paths")) {
        if (value < 1024 || value > (1ll << 31)) return fail(c, MIRT_ERR_ARG, "max_paths must be in [1024, 2^31]");
        c->max_paths = (uint64_t)value;
        return MIRT_OK;
    }
    if (!strcmp(name, "time_launches")) { c->time_launches = value != 0; return MIRT_OK; }
    if (!strcmp(name, "lds_budget")) { if (value < 4096 || value > 64 * 1024) return fail(c, MIRT_ERR_ARG, "lds_budget out of range"); c->lds_budget = (uint32_t)value; return MIRT_OK; }
    if (!strcmp(name, "engine")) { if (value < 0 || value > 1) return fail(c, MIRT_ERR_ARG, "engine must be 0 (wavefront) or 1 (persistent)"); c->engine = (uint32_t)value; return MIRT_OK; }
    if (!strcmp(name, "lds_prims")) { if (value < 0 || value > 64 * 1024) return fail(c, MIRT_ERR_ARG, "lds_prims out of range"); c->lds_prims = (uint32_t)value; return MIRT_OK; }
    if (!strcmp(name, "lds_mats")) { if (value < 0 || value > 64 * 1024) return fail(c, MIRT_ERR_ARG, "lds_mats out of range"); c->lds_mats = (uint32_t)value; return MIRT_OK; }
    if (!strcmp(name, "stack_lds")) { if (value < 1 || value > 64) return fail(c, MIRT_ERR_ARG, "stack_lds out of range"); c->stack_lds = (uint32_t)value; return MIRT_OK; }
    if (!strcmp(name, "leaf_wait")) { if (value == 0 || value < -1 || value > 64) return fail(c, MIRT_ERR_ARG, "leaf_wait out of range"); c->leaf_wait = (int)value; return MIRT_OK; }
    if (!strcmp(name, "lds_top")) { if (value < -1 || value > 256) return fail(c, MIRT_ERR_ARG, "lds_top out of range"); c->lds_top = (int32_t)value; return MIRT_OK; }
    if (!strcmp(name, "walk_yield")) { if (value < -1 || value > 64) return fail(c, MIRT_ERR_ARG, "walk_yield out of range"); c->walk_yield = (int32_t)value; return MIRT_OK; }
    if (!strcmp(name, "tile_paths")) { c->tile_paths = value ? 1u : 0u; return MIRT_OK; }
    if (!strcmp(name, "paths_per_lane")) { if (value != 1 && value != 2) return fail(c, MIRT_ERR_ARG, "paths_per_lane is 1 or 2"); c->paths_per_lane = (uint32_t)value; return MIRT_OK; }
    if (!strcmp(name, "blocks_per_cu")) { if (value < 0 || value > 65536) return fail(c, MIRT_ERR_ARG, "blocks_per_cu out of range"); c->blocks_per_cu = (uint32_t)value; return MIRT_OK; }
    return fail(c, MIRT_ERR_ARG, std::string("unknown option ") + name);
}

// ------------------------------------------------------------------------------------------ scene
extern "C" int mirt_upload_scene(mirt_ctx *c, const mirt_scene *s)
{
    if (!c || !s) return MIRT_ERR_ARG;
    CHECK_SIZE(c, s, mirt_scene);
    HIPCHK(c, hipSetDevice(c->device));
    std::string why = host::validate(*s);
    if (!why.empty()) return fail(c, MIRT_ERR_ARG, "mirt_upload_scene: " + why);
    host::Tree tree;
    why = host::prepare_tree(*s, tree);
    if (!why.empty()) return fail(c, MIRT_ERR_ARG, "mirt_upload_scene: " + why);
    if (tree.depth > 60) return fail(c, MIRT_ERR_ARG, "mirt_upload_scene: BVH deeper than 60 levels");
    if (tree.box.size() >= (1u << 25) || s->n_objects >= (1u << 26)) return fail(c, MIRT_ERR_ARG, "mirt_upload_scene: 2^25 BVH nodes or 2^26 objects, or more (the walks address them by 32-bit byte offsets)");
    // From here on the context's buffers are rewritten in place (DevBuf::ensure frees before it allocates): until every
    // allocation and copy has succeeded the context has no scene, so a failed re-upload cannot leave render or batch
    // calls with pointers to freed buffers.
    c->have_scene = false;

    std::vector<PrimRec> prims(s->n_prims);
    bool has_chains = false;
    for (uint32_t p = 0; p < s->n_prims; ++p) {
        PrimRec &r = prims[p];
        memset(&r, 0, sizeof r);
        for (int i = 0; i < 3; ++i) r.off[i] = s->prim_off[3 * (size_t)p + i];
        const float *g = s->prim_geom + 5 * (size_t)p;
        r.mat = s->prim_mat[p];
        const bool chain = host::chain_form(*s, p);                        // a leaf with its own list of instances (rt_device.h OpRec)
        if (chain) {
            has_chains = true;
            const bool sphere = s->prim_kind[p] == MIRT_PRIM_SPHERE;
            for (int i = 0; i < 3; ++i) r.off[i] = sphere ? s->prim_center[3 * (size_t)p + i] : 0.0f;
            const uint32_t ax = s->prim_axes[p];
            r.info = (sphere ? 0u : (1u | ((ax & 3) << 1) | (((ax >> 2) & 3) << 3) | (((ax >> 4) & 3) << 5) | (((ax >> 6) & 1) << 7))) | (1u << 9) | (s->prim_nops[p] << 16);
            if (sphere) { r.ga[0] = g[0]; r.ga[1] = 1.f / g[0]; r.ga[2] = g[0] * g[0]; }
            else { r.ga[0] = g[0]; r.ga[1] = g[1]; r.ga[2] = g[2]; r.gb[0] = g[3]; r.gb[1] = g[4]; }
            const uint32_t first = s->prim_chain[p] - 1u;
            memcpy(&r.rot[0], &first, 4);
        } else if (s->prim_kind[p] == MIRT_PRIM_SPHERE) {
            r.info = 0;
            r.ga[0] = g[0]; r.ga[1] = 1.f / g[0]; r.ga[2] = g[0] * g[0];   // Vec3.h:85-88, Hitables.cpp:23
        } else {
            uint32_t ax = s->prim_axes[p];
            const float *q = s->prim_rot + 6 * (size_t)p;
            bool chain = (ax >> 7) & 1, live = false;
            if (chain) for (int k = 0; k < 3; ++k) live |= !(q[2 * k] == 0.0f && q[2 * k + 1] == 1.0f);
            r.info = 1u | ((ax & 3) << 1) | (((ax >> 2) & 3) << 3) | (((ax >> 4) & 3) << 5) | (((ax >> 6) & 1) << 7) | ((live ? 1u : 0u) << 8);
            r.ga[0] = g[0]; r.ga[1] = g[1]; r.ga[2] = g[2]; r.gb[0] = g[3]; r.gb[1] = g[4];
            for (int i = 0; i < 6; ++i) r.rot[i] = chain ? q[i] : ((i & 1) ? 1.0f : 0.0f);
        }
    }
    // The list walk's copy: stable-sorted by class so that each class is one tight loop (rt_kernels.h closest_list);
    // .mat carries the primitive's own index, which is what acceptance and the caller need.
    auto list_class = [](const PrimRec &r) -> uint32_t {
        if (PI_ROT(r.info) || PI_CHAIN(r.info)) return 4u;
        if (PI_KIND(r.info) == MIRT_PRIM_SPHERE) return 3u;
        const uint32_t a = PI_AAX(r.info), b = PI_BAX(r.info), cc = PI_CAX(r.info);
        if (a == 0 && b == 1 && cc == 2) return 0u;
        if (a == 0 && b == 2 && cc == 1) return 1u;
        if (a == 2 && b == 1 && cc == 0) return 2u;
        return 4u;
    };
    std::vector<PrimRec> lprims; lprims.reserve(s->n_prims);
    uint32_t lend[4] = { 0, 0, 0, 0 };
    for (uint32_t cls = 0; cls < 5; ++cls) {
        for (uint32_t p = 0; p < s->n_prims; ++p)
            if (list_class(prims[p]) == cls) { lprims.push_back(prims[p]); lprims.back().mat = p; }
        if (cls < 4) lend[cls] = (uint32_t)lprims.size();
    }
    // Facing pairs (rt_kernels.h rect_pair): in each orientation class the rects with the lowest and the highest plane move to the
    // front of the class's run, low one first.  The plane is k + offset along the normal; which two are picked decides only how
    // often the merged test applies, never a result (the kernel checks "between" per ray in the test's own arithmetic).
    uint32_t pair_mask = 0;
    for (uint32_t cls = 0; cls < 3; ++cls) {
        const uint32_t first = cls ? lend[cls - 1] : 0u, end = lend[cls];
        if (end - first < 2u) continue;
        const uint32_t cax = cls == 0 ? 2u : (cls == 1 ? 1u : 0u);
        auto plane = [&](uint32_t i) { return (double)lprims[i].gb[1] + (double)lprims[i].off[cax]; };
        uint32_t lo = first, hi = first;
        for (uint32_t i = first; i < end; ++i) { if (plane(i) < plane(lo)) lo = i; if (plane(i) > plane(hi)) hi = i; }
        if (!(plane(lo) < plane(hi))) continue;                       // all in one plane (or NaN): no pair
        std::swap(lprims[first], lprims[lo]);
        if (hi == first) hi = lo;                                      // the high one was sitting where the low one went
        std::swap(lprims[first + 1], lprims[hi]);
        pair_mask |= 1u << cls;
        // what rect_pair reads of both, in rotation words that unrotated records do not use (rot[2..5] = the record's last 16 bytes)
        PrimRec &a = lprims[first], &b = lprims[first + 1];
        a.rot[2] = a.gb[1]; a.rot[3] = a.off[cax]; a.rot[4] = b.gb[1]; a.rot[5] = b.off[cax];
        memcpy(&b.rot[2], &a.mat, 4); memcpy(&b.rot[3], &b.mat, 4);
    }
    std::vector<NodeRec> nodes(tree.box.size());
    for (size_t i = 0; i < nodes.size(); ++i) {
        for (int a = 0; a < 3; ++a) { nodes[i].mn[a] = tree.box[i].mn[a]; nodes[i].mx[a] = tree.box[i].mx[a]; }
        nodes[i].left = tree.left[i]; nodes[i].right = tree.right[i];
    }
    std::vector<Node2> nodes2(tree.box.size());
    for (size_t i = 0; i < nodes2.size(); ++i) {
        Node2 &n = nodes2[i]; memset(&n, 0, sizeof n);
        n.left = tree.left[i]; n.right = tree.right[i];
        n.lleaf = (uint32_t)n.left >> 31; n.rleaf = (uint32_t)n.right >> 31;
        auto child_box = [&](int32_t ch) -> host::Box {
            if (ch == MIRT_CHILD_NONE) { host::Box e; for (int a = 0; a < 3; ++a) { e.mn[a] = 3.402823466e+38f; e.mx[a] = -3.402823466e+38f; } return e; }
            return ch < 0 ? tree.obox[~ch] : tree.box[ch];
        };
        host::Box l = child_box(n.left), r = child_box(n.right);
        for (int a = 0; a < 3; ++a) { n.lmn[a] = l.mn[a]; n.lmx[a] = l.mx[a]; n.rmn[a] = r.mn[a]; n.rmx[a] = r.mx[a]; }
        if (n.left >= 0) n.left <<= 6;                 // children that are nodes as byte offsets (fetch_node2): no shift per step
        if (n.right >= 0) n.right <<= 6;
    }
    std::vector<uint2> objs(s->n_objects);
    bool single = true;
    for (uint32_t o = 0; o < s->n_objects; ++o) { objs[o] = make_uint2(s->obj_first[o], s->obj_count[o]); single &= s->obj_count[o] == 1; }
    std::vector<MatRec> mats(s->n_mats);
    for (uint32_t m = 0; m < s->n_mats; ++m) {
        MatRec &r = mats[m]; memset(&r, 0, sizeof r);
        r.kind = s->mat_kind[m]; r.fuzz = s->mat_fuzz[m]; r.ior = s->mat_ior[m];
        r.tex = (s->n_textures && s->mat_texture[m] >= 0) ? s->mat_texture[m] : -1;
        r.inv_ior = 1.0f / r.ior;
        r.r0 = schlick_r0(r.ior);                       // same IEEE divide and multiply as in the kernel (no contraction on either side)
        for (int i = 0; i < 3; ++i) { r.albedo[i] = s->mat_albedo[3 * (size_t)m + i]; r.emit[i] = s->mat_emit[3 * (size_t)m + i]; }
    }
    if (c->d_prims.ensure(prims.size() * sizeof(PrimRec)) || c->d_lprims.ensure(lprims.size() * sizeof(PrimRec)) || c->d_nodes.ensure(nodes.size() * sizeof(NodeRec)) ||
        c->d_nodes2.ensure(nodes2.size() * sizeof(Node2)) ||
        c->d_objs.ensure(objs.size() * sizeof(uint2)) || c->d_mats.ensure(mats.size() * sizeof(MatRec)))
        return fail(c, MIRT_ERR_NOMEM, "mirt_upload_scene: hipMalloc failed");
    HIPCHK(c, hipMemcpy(c->d_prims.p, prims.data(), prims.size() * sizeof(PrimRec), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_lprims.p, lprims.data(), lprims.size() * sizeof(PrimRec), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_nodes.p, nodes.data(), nodes.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
#if MIRT_NODE_SOA
    {   // measurement build: the same records as sixteen planes of one dword per node
        std::vector<uint32_t> planes(nodes2.size() * 16);
        for (size_t i = 0; i < nodes2.size(); ++i) for (size_t k = 0; k < 16; ++k) planes[k * nodes2.size() + i] = ((const uint32_t *)&nodes2[i])[k];
        HIPCHK(c, hipMemcpy(c->d_nodes2.p, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
    }
#else
    HIPCHK(c, hipMemcpy(c->d_nodes2.p, nodes2.data(), nodes2.size() * sizeof(Node2), hipMemcpyHostToDevice));
#endif
    HIPCHK(c, hipMemcpy(c->d_objs.p, objs.data(), objs.size() * sizeof(uint2), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_mats.p, mats.data(), mats.size() * sizeof(MatRec), hipMemcpyHostToDevice));

    SceneDev &d = c->sc;
    d.texs = nullptr; d.texels = nullptr; d.n_textures = s->n_textures;
    if (s->n_textures) {
        std::vector<TexRec> texs(s->n_textures);
        uint64_t n_texels = 0;
        for (uint32_t t = 0; t < s->n_textures; ++t) {
            texs[t] = TexRec{ s->tex_width[t], s->tex_height[t], s->tex_offset[t], 0u };
            n_texels = std::max<uint64_t>(n_texels, (uint64_t)s->tex_offset[t] + (uint64_t)s->tex_width[t] * s->tex_height[t]);
        }
        if (c->d_texs.ensure(texs.size() * sizeof(TexRec)) || c->d_texels.ensure(n_texels * 4)) return fail(c, MIRT_ERR_NOMEM, "mirt_upload_scene: hipMalloc failed");
        HIPCHK(c, hipMemcpy(c->d_texs.p, texs.data(), texs.size() * sizeof(TexRec), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_texels.p, s->tex_rgba8, n_texels * 4, hipMemcpyHostToDevice));   // R,G,B,A bytes = one little-endian word per texel
        d.texs = (const TexRec *)c->d_texs.p; d.texels = (const uint32_t *)c->d_texels.p;
    }
    d.prims = (const PrimRec *)c->d_prims.p; d.n_prims = s->n_prims;
    d.lprims = (const PrimRec *)c->d_lprims.p; for (int i = 0; i < 4; ++i) d.lend[i] = lend[i];
    d.list_fast = 1u;
    for (uint32_t i = 0; i < lend[2]; ++i) {
        const PrimRec &r = lprims[i];
        if (!(std::fabs(r.gb[1]) <= 0x1p38f && std::fabs(r.off[0]) <= 0x1p38f && std::fabs(r.off[1]) <= 0x1p38f && std::fabs(r.off[2]) <= 0x1p38f)) d.list_fast = 0u;
    }
    if (d.list_fast) d.list_fast |= pair_mask << 1;
    d.nodes = (const NodeRec *)c->d_nodes.p; d.n_nodes = (uint32_t)nodes.size();
    d.nodes2 = (const Node2 *)c->d_nodes2.p; d.single = single ? 1u : 0u;
    d.objs = (const uint2 *)c->d_objs.p; d.n_objects = s->n_objects;
    // scenes whose objects are all single unrotated spheres: the 16 bytes a leaf test reads, per object (rt_kernels.h sphere_leaf_t)
    d.sph4 = nullptr;
    bool all_spheres = single && s->n_prims == s->n_objects && s->n_prims > 0;
    for (uint32_t p = 0; all_spheres && p < s->n_prims; ++p) all_spheres = prims[p].info == 0u && objs[p].x == p;
    if (all_spheres) {
        std::vector<float4> sph(s->n_prims);
        for (uint32_t p = 0; p < s->n_prims; ++p) sph[p] = make_float4(prims[p].off[0], prims[p].off[1], prims[p].off[2], prims[p].ga[2]);
        if (c->d_sph4.ensure(sph.size() * sizeof(float4))) return fail(c, MIRT_ERR_NOMEM, "mirt_upload_scene: hipMalloc failed");
        HIPCHK(c, hipMemcpy(c->d_sph4.p, sph.data(), sph.size() * sizeof(float4), hipMemcpyHostToDevice));
        d.sph4 = (const float4 *)c->d_sph4.p;
    } else c->d_sph4.release();                                   // a scene that cannot use the records does not keep the last one's
    d.ops = nullptr; c->has_chains = has_chains;
    if (has_chains && s->n_ops) {
        std::vector<OpRec> ops(s->n_ops);
        for (uint32_t k = 0; k < s->n_ops; ++k) ops[k] = OpRec{ s->op_kind[k], s->op_arg[3 * (size_t)k], s->op_arg[3 * (size_t)k + 1], s->op_arg[3 * (size_t)k + 2] };
        if (c->d_ops.ensure(ops.size() * sizeof(OpRec))) return fail(c, MIRT_ERR_NOMEM, "mirt_upload_scene: hipMalloc failed");
        HIPCHK(c, hipMemcpy(c->d_ops.p, ops.data(), ops.size() * sizeof(OpRec), hipMemcpyHostToDevice));
        d.ops = (const OpRec *)c->d_ops.p;
    }
    d.mats = (const MatRec *)c->d_mats.p; d.n_mats = s->n_mats;
    d.root = tree.root;
    for (int i = 0; i < 3; ++i) d.ambient[i] = s->ambient[i];
    // LDS plan: the short part of the traversal stack, the material table, small primitive tables, and whatever
    // is left of the budget for the top of the tree.  Occupancy beats LDS-resident nodes (measured: 1M spheres,
    // 40 KiB plan 2.07 Grays/s, 8 KiB plan 2.65), so the default budget is small.
    const uint32_t budget = c->lds_budget;
    d.stack_depth = tree.depth + 1;
    d.stack_lds = std::min<uint32_t>(d.stack_depth, c->stack_lds);
    d.spill = nullptr; d.spill_stride = 0;
    uint32_t used = d.stack_lds * MIRT_BLOCK * 4u;
    d.lds_mats = (s->n_mats * sizeof(MatRec) <= c->lds_mats) ? s->n_mats : 0u; used += d.lds_mats * (uint32_t)sizeof(MatRec);
    c->small_scene = (size_t)s->n_prims * sizeof(PrimRec) <= c->lds_prims;       // scenes the list walk is meant for; larger ones keep the LDS for residency
    c->prims_in_lds = c->small_scene && d.lds_mats == s->n_mats;                 // a long material table (it need not be used) leaves both in HBM
    d.lds_prims = c->prims_in_lds ? s->n_prims : 0u; used += d.lds_prims * MIRT_LDS_PRIM_STRIDE;
    const uint32_t room_bytes = used < budget ? budget - used : 0u;
    c->sc2 = d;
    d.node_bytes = (uint32_t)sizeof(NodeRec);
    d.lds_nodes = std::min<uint32_t>(room_bytes / d.node_bytes, d.n_nodes);
    c->lds_bytes = used + d.lds_nodes * d.node_bytes;
    c->lds_stack_bytes = d.stack_lds * MIRT_BLOCK * 4u;
    c->sc2.node_bytes = (uint32_t)sizeof(Node2);
    if (c->sc2.root >= 0) c->sc2.root <<= 6;                        // the near-first walks name a node by its byte offset (Node2::left)
    // near-first walks read their nodes from memory, except the sliced walk of a large tree, which takes the top levels from LDS
    // (k_trace_bvh<4, ...>, rt_kernels.h bvh2_slice<.., LTOP>): option lds_top, -1 = 32 nodes for trees of 65536 nodes and more
    c->sc2.lds_nodes = std::min<uint32_t>(c->lds_top >= 0 ? (uint32_t)c->lds_top : (c->sc2.n_nodes >= 65536u ? 32u : 0u), c->sc2.n_nodes);
#if MIRT_NODE_SOA
    c->sc2.lds_nodes = 0;                                           // measurement build: the planes are not records, nothing to stage
#endif
    c->lds_bytes2 = used + c->sc2.lds_nodes * c->sc2.node_bytes;
    c->have_scene = true;
    return MIRT_OK;
}

// Host-only (no GPU, no context): the library's own BVH builder, the one mirt_upload_scene runs when a scene
// comes without a tree.  Nodes come out numbered breadth-first, in the node_* format of mirt_scene.
extern "C" int mirt_build_bvh(const mirt_scene *scene, uint32_t capacity_nodes, float *node_min, float *node_max,
                              int32_t *node_left, int32_t *node_right, int32_t *root, uint32_t *n_nodes, uint32_t *depth)
{
    if (!scene || !node_min || !node_max || !node_left || !node_right || !root || !n_nodes) return MIRT_ERR_ARG;
    CHECK_SIZE(nullptr, scene, mirt_scene);
    mirt_scene s = *scene; s.n_nodes = 0;
    if (!host::validate(s).empty()) return MIRT_ERR_ARG;
    host::Tree tree;
    if (!host::prepare_tree(s, tree).empty()) return MIRT_ERR_ARG;
    if (tree.box.size() > capacity_nodes) return MIRT_ERR_NOMEM;
    for (size_t i = 0; i < tree.box.size(); ++i) {
        for (int a = 0; a < 3; ++a) { node_min[3 * i + a] = tree.box[i].mn[a]; node_max[3 * i + a] = tree.box[i].mx[a]; }
        node_left[i] = tree.left[i]; node_right[i] = tree.right[i];
    }
    *root = tree.root; *n_nodes = (uint32_t)tree.box.size();
    if (depth) *depth = tree.depth;
    return MIRT_OK;
}

extern "C" int mirt_set_camera(mirt_ctx *c, const mirt_camera *cam)
{
    if (!c || !cam) return MIRT_ERR_ARG;
    c->cam = *cam; c->have_camera = true;
    return MIRT_OK;
}

// ------------------------------------------------------------------------------------------ render
// 0 = list, 1 = BVH walked in the reference's order, 2 = BVH near-first
static int accel_mode(const mirt_ctx *c, uint32_t flags)
{
    if (flags & MIRT_FLAG_ACCEL_LIST) return 0;
    if (flags & MIRT_FLAG_ACCEL_AUTO) return (c->sc.n_prims <= 32 && c->small_scene) ? 0 : 2;   // tools/crossover.py: the list wins up to ~30 primitives
    if (flags & MIRT_FLAG_ACCEL_ORDERED) return 2;
    return 1;
}

// Points sd's spill area at a buffer large enough for `grid` workgroups.
static int bind_spill(mirt_ctx *c, SceneDev &sd, uint32_t grid)
{
    const uint32_t levels = sd.stack_depth - sd.stack_lds;
    sd.spill_stride = grid * MIRT_BLOCK;
    if (levels && c->d_spill.ensure((size_t)levels * sd.spill_stride * sizeof(int32_t))) return -1;
    sd.spill = (int32_t *)c->d_spill.p;
    return 0;
}

template <bool FIRST>
static void launch_bounce(const mirt_ctx *c, int mode, const BounceArgs &a, uint32_t grid, hipStream_t st)
{
    const uint32_t lds = mode == 2 ? c->lds_bytes2 : (mode == 0 ? c->lds_bytes - c->lds_stack_bytes : c->lds_bytes);
    const bool pl = c->prims_in_lds;
    dim3 g(grid), b(MIRT_BLOCK);
    if (mode == 0) { if (pl) hipLaunchKernelGGL((k_bounce<FIRST, 0, true>), g, b, lds, st, a); else hipLaunchKernelGGL((k_bounce<FIRST, 0, false>), g, b, lds, st, a); }
    else if (mode == 1) { if (pl) hipLaunchKernelGGL((k_bounce<FIRST, 1, true>), g, b, lds, st, a); else hipLaunchKernelGGL((k_bounce<FIRST, 1, false>), g, b, lds, st, a); }
    else { if (pl) hipLaunchKernelGGL((k_bounce<FIRST, 2, true>), g, b, lds, st, a); else hipLaunchKernelGGL((k_bounce<FIRST, 2, false>), g, b, lds, st, a); }
}

// Workgroups of the depth>=1 kernel that fit on one CU with this scene's LDS plan (the persistent grid is sized to it).
static uint32_t bounce_occupancy(const mirt_ctx *c, int mode)
{
    const uint32_t lds = mode == 2 ? c->lds_bytes2 : (mode == 0 ? c->lds_bytes - c->lds_stack_bytes : c->lds_bytes);
    const bool pl = c->prims_in_lds;
    int n = 0;
    hipError_t e;
    if (mode == 0) e = pl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 0, true>, MIRT_BLOCK, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 0, false>, MIRT_BLOCK, lds);
    else if (mode == 1) e = pl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 1, true>, MIRT_BLOCK, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 1, false>, MIRT_BLOCK, lds);
    else e = pl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 2, true>, MIRT_BLOCK, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, 2, false>, MIRT_BLOCK, lds);
    if (e != hipSuccess || n < 1) { (void)hipGetLastError(); return 8; }
    return (uint32_t)n;
}

// k_trace<PLDS, FULL> (list) or k_trace_bvh<ACCEL, PLDS, FULL> for this scene and mode (FULL: image textures and/or chain-form primitives)
#define MIRT_TRACE_DISPATCH(mode, pl, tex, CALL)                                                                  \
    do {                                                                                                          \
        if (mode == 0) { if (pl) { if (tex) { CALL(0, true, true); } else { CALL(0, true, false); } } else { if (tex) { CALL(0, false, true); } else { CALL(0, false, false); } } }      \
        else if (mode == 1) { if (pl) { if (tex) { CALL(1, true, true); } else { CALL(1, true, false); } } else { if (tex) { CALL(1, false, true); } else { CALL(1, false, false); } } } \
        else if (mode == 2) { if (pl) { if (tex) { CALL(2, true, true); } else { CALL(2, true, false); } } else { if (tex) { CALL(2, false, true); } else { CALL(2, false, false); } } } \
        else if (mode == 3) { if (pl) { if (tex) { CALL(3, true, true); } else { CALL(3, true, false); } } else { if (tex) { CALL(3, false, true); } else { CALL(3, false, false); } } } \
        else { if (pl) { if (tex) { CALL(4, true, true); } else { CALL(4, true, false); } } else { if (tex) { CALL(4, false, true); } else { CALL(4, false, false); } } }                \
    } while (0)

static uint32_t trace_lds(const mirt_ctx *c, int mode)
{
    return mode >= 2 ? c->lds_bytes2 : (mode == 0 ? c->lds_bytes - c->lds_stack_bytes : c->lds_bytes);
}
// The list walk of a scene without image textures runs two paths per lane (k_trace2) when the paths_per_lane option says 2
static bool full_variant(const mirt_ctx *c) { return c->sc.n_textures != 0 || c->has_chains; }   // image textures and/or chain-form primitives
static bool two_paths(const mirt_ctx *c, int mode) { return mode == 0 && !full_variant(c) && c->paths_per_lane == 2 && !MIRT_DEFER_MD; }
static void launch_trace(const mirt_ctx *c, int mode, const TraceArgs &a, uint32_t grid, hipStream_t st)
{
    const uint32_t lds = trace_lds(c, mode);
    const bool pl = c->prims_in_lds, tex = full_variant(c);
    dim3 g(grid), b(MIRT_BLOCK);
    if (two_paths(c, mode)) { if (pl) hipLaunchKernelGGL((k_trace2<true>), g, b, lds, st, a); else hipLaunchKernelGGL((k_trace2<false>), g, b, lds, st, a); return; }
#define MIRT_CALL(A, P, T) do { if (A == 0) hipLaunchKernelGGL((k_trace<P, T>), g, b, lds, st, a); else hipLaunchKernelGGL((k_trace_bvh<(A == 0 ? 1 : A), P, T>), g, b, lds, st, a); } while (0)
    MIRT_TRACE_DISPATCH(mode, pl, tex, MIRT_CALL);
#undef MIRT_CALL
}
static uint32_t trace_occupancy(const mirt_ctx *c, int mode)
{
    const uint32_t lds = trace_lds(c, mode);
    const bool pl = c->prims_in_lds, tex = full_variant(c);
    int n = 0;
    hipError_t e = hipSuccess;
    if (two_paths(c, mode)) {
        e = pl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace2<true>, MIRT_BLOCK, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace2<false>, MIRT_BLOCK, lds);
        if (e != hipSuccess || n < 1) { (void)hipGetLastError(); return 5; }
        return (uint32_t)n;
    }
#define MIRT_CALL(A, P, T) do { if (A == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace<P, T>, MIRT_BLOCK, lds); else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_bvh<(A == 0 ? 1 : A), P, T>, MIRT_BLOCK, lds); } while (0)
    MIRT_TRACE_DISPATCH(mode, pl, tex, MIRT_CALL);
#undef MIRT_CALL
    if (e != hipSuccess || n < 1) { (void)hipGetLastError(); return 8; }
    return (uint32_t)n;
}

// The persistent engine: one k_trace launch + one k_accumulate per batch; no queues.
// fixed: the scratch of mirt_render_device_noalloc (nothing is allocated, sizes beyond it are errors); NULL: the
// context's own buffers, grown on demand.
static int render_persistent(mirt_ctx *c, const mirt_render_params *p, float *d_sum, hipStream_t st, mirt_stats *stats,
                             int mode, uint32_t spp_begin, uint32_t spp_count, bool one_spp, const Workspace *fixed = nullptr)
{
    const uint64_t npix_total = (uint64_t)p->width * p->height;
    uint64_t cap = std::min<uint64_t>(c->max_paths, npix_total * spp_count);
    if (fixed) cap = std::min<uint64_t>(cap, fixed->rad_paths);
    // -1 (default) = 56 for batches of 2 Mi paths or more: with the resumable walk compiled for 8 waves per SIMD like the plain
    // one (its state costs 3-4 dwords of scratch) the slices pay from ~4 M paths on (485 spheres 1200x800@500 +13 %, 1 M spheres
    // 1920x1080@128 +10 %; 52/56/60/62 lanes: within 1 % / best / -1 % / -7 %); launches of ~1 M paths and fewer are 4 % faster plain.
    // Round 3 swept both thresholds again on the final kernels (tools/sweep_walk.sh): a small tree (no top levels in LDS), whose
    // walk is bound by instruction issue, is 1.2 % faster with 52 finished lanes and leaf passes of 8 waiting lanes; the 1 M-sphere
    // tree, whose walk waits on memory, is flat around 56 / 4
    const bool small_tree = c->sc2.lds_nodes == 0;
    const uint32_t yield_want = c->walk_yield > 0 ? (uint32_t)c->walk_yield : ((c->walk_yield < 0 && cap >= (2ull << 20)) ? (small_tree ? 52u : 56u) : 0u);
    if (mode == 2 && yield_want) mode = small_tree ? 3 : 4;          // the near-first walk in resumable slices (k_trace_bvh<3, ...>; <4, ...>: top of the tree from LDS)
    const uint32_t npix_b = (uint32_t)std::min<uint64_t>(npix_total, cap);
    const uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spp_count, cap / npix_b));
    const size_t capq = (size_t)npix_b * S;
    const uint32_t n_pix_tiles = (uint32_t)((npix_total + npix_b - 1) / npix_b);
    const uint32_t n_s_tiles = (spp_count + S - 1) / S;
    const uint64_t n_batches64 = (uint64_t)n_pix_tiles * n_s_tiles;
    const size_t per_batch = 66;                           // cursor, hist[65]
    if (n_batches64 > 0x7FFFFFFFull / per_batch) return fail(c, MIRT_ERR_ARG, "mirt_render_device: too many batches (raise the max_paths option)");
    const uint32_t n_batches = (uint32_t)n_batches64;
    const uint32_t per_cu = c->blocks_per_cu ? c->blocks_per_cu : trace_occupancy(c, mode);
    if (getenv("MIRT_DEBUG")) fprintf(stderr, "[mirt] persistent engine, mode %d, lds %u B, %u workgroups/CU (occupancy query %u)\n", mode, trace_lds(c, mode), per_cu, trace_occupancy(c, mode));
    const uint32_t max_grid = (uint32_t)c->n_cu * per_cu;
    TraceArgs a{};
    a.sc = mode >= 2 ? c->sc2 : c->sc; a.cam = c->cam;
    a.slice_want = yield_want; a.leaf_want = c->leaf_wait > 0 ? (uint32_t)c->leaf_wait : (mode == 3 ? 8u : MIRT_LEAF_WANT);
    uint32_t *counts;
    if (fixed) {
        // nothing is allocated on this path: every size is checked against the fixed scratch instead
        const size_t spill_levels = a.sc.stack_depth - a.sc.stack_lds;
        if (n_batches * per_batch * sizeof(uint32_t) > fixed->counts_bytes)
            return fail(c, MIRT_ERR_ARG, "mirt_render_device_noalloc: width*height*spp_count needs " + std::to_string(n_batches) + " batches of 2 Mi paths; the fixed scratch holds the counters of " +
                                         std::to_string(fixed->counts_bytes / (per_batch * sizeof(uint32_t))) + " (render fewer samples per call)");
        if (spill_levels * max_grid * MIRT_BLOCK * sizeof(int32_t) > fixed->spill_bytes)
            return fail(c, MIRT_ERR_ARG, "mirt_render_device_noalloc: the BVH is " + std::to_string(a.sc.stack_depth - 1) + " levels deep and its traversal stacks need " +
                                         std::to_string(spill_levels * max_grid * MIRT_BLOCK * sizeof(int32_t)) + " bytes of scratch; the fixed scratch has " + std::to_string(fixed->spill_bytes) +
                                         " (use mirt_render_device, or lower the blocks_per_cu option)");
        a.rad = fixed->rad; counts = fixed->counts;
        a.sc.spill = fixed->spill; a.sc.spill_stride = max_grid * MIRT_BLOCK;
    } else {
        if (c->d_rad.ensure(capq * sizeof(RadSlot)) || c->d_counts.ensure(n_batches * per_batch * sizeof(uint32_t)))
            return fail(c, MIRT_ERR_NOMEM, "mirt_render_device: hipMalloc of the radiance slots failed (lower the max_paths option)");
        a.rad = (RadSlot *)c->d_rad.p; counts = (uint32_t *)c->d_counts.p;
        if (bind_spill(c, a.sc, max_grid)) return fail(c, MIRT_ERR_NOMEM, "mirt_render_device: hipMalloc of the traversal-stack spill area failed");
    }
    HIPCHK(c, hipMemsetAsync(counts, 0, n_batches * per_batch * sizeof(uint32_t), st));
#if MIRT_DEFER_MD
    if (mode == 0 && !full_variant(c)) {                             // the second stream's per-wave queues (experimental build)
        if (fixed) return fail(c, MIRT_ERR_STATE, "mirt_render_device_noalloc: not available in a MIRT_DEFER_MD build");
        const size_t waves = (size_t)max_grid * (MIRT_BLOCK / 64);
        if (c->d_mdq.ensure(waves * MDQ_PLANES * MDQ_CAP * sizeof(float4)) || c->d_rdy.ensure(waves * RDY_PLANES * RDY_CAP * sizeof(float4)))
            return fail(c, MIRT_ERR_NOMEM, "mirt_render_device: hipMalloc of the second stream's queues failed");
        a.mdq = (float4 *)c->d_mdq.p; a.rdy = (float4 *)c->d_rdy.p;
    }
#endif
#if (MIRT_TEXP & 16)
    {   // schedule dump build: two float4 per wave of the grid (trace_body), written to $MIRT_WAVE_DUMP after the render
        const size_t waves = (size_t)max_grid * (MIRT_BLOCK / 64);
        if (c->d_mdq.ensure(waves * 2 * sizeof(float4))) return fail(c, MIRT_ERR_NOMEM, "wave dump buffer");
        HIPCHK(c, hipMemsetAsync(c->d_mdq.p, 0, waves * 2 * sizeof(float4), st));
        a.mdq = (float4 *)c->d_mdq.p;
    }
#endif
    if (stats) HIPCHK(c, hipEventRecord(c->ev0, st));
    a.width = p->width; a.height = p->height; a.max_depth = p->max_depth; a.flags = p->flags;
    a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
    // BVH walks gain ~2 % from compact 8x8 patches of camera rays; the list walk loses ~3 % (paths of one patch end together)
    a.tiles_x = (c->tile_paths && mode != 0 && p->width % 8u == 0 && p->height % 8u == 0) ? p->width / 8u : 0u;
    a.by_width = fastdiv_make(a.width); a.by_tiles_x = fastdiv_make(a.tiles_x);
    uint32_t batch = 0;
    const bool timed = stats && c->time_launches;
    if (timed) while (c->ev_pool.size() < 2 * (size_t)n_batches) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->ev_pool.push_back(e); }
    for (uint32_t pt = 0; pt < n_pix_tiles; ++pt) {
        const uint32_t pix0 = pt * npix_b;
        const uint32_t np = (uint32_t)std::min<uint64_t>(npix_b, npix_total - pix0);
        for (uint32_t stile = 0; stile < n_s_tiles; ++stile, ++batch) {
            const uint32_t s0 = spp_begin + stile * S;
            const uint32_t ns = std::min<uint32_t>(S, spp_begin + spp_count - s0);
            a.pix0 = pix0; a.npix = np; a.spp0 = s0; a.n_paths = np * ns; a.by_npix = fastdiv_make(np);
            a.cursor = counts + batch * per_batch; a.hist = a.cursor + 1;
            // A small launch runs on fewer workgroups per CU than fit: with only a few paths per lane the launch is as long as its
            // longest chain of bounces, and every wave of a fuller SIMD advances more slowly.  Measured (profiles/r02_small_launches.txt):
            // list walk, up to ~1.2 M paths: two or three workgroups per CU, about six paths per resident lane (Cornell 128^2@16
            // 0.34 -> 0.24 ms, 1024^2 at 1 spp 0.51 -> 0.37 ms; between 1.2 M and 8 M paths the grid size measures within noise);
            // BVH walks: four up to 512 k paths, five up to 4 M (485 spheres 1024^2 at 1 spp 0.92 -> 0.79 ms); the whole CU beyond.
            uint32_t grid_cap = max_grid;
            const uint64_t per_share = (uint64_t)c->n_cu * MIRT_BLOCK * 6;
            if (!c->blocks_per_cu) {
                uint64_t want = per_cu;
                if (mode == 0) { if (a.n_paths <= 3 * per_share) want = std::max<uint64_t>(2, (a.n_paths + per_share - 1) / per_share); }
                else if (a.n_paths <= (512u << 10)) want = 4;
                else if (a.n_paths <= (4u << 20)) want = 5;
                grid_cap = (uint32_t)c->n_cu * (uint32_t)std::min<uint64_t>(per_cu, want);
            }
            const uint64_t waves = (uint64_t)grid_cap * (MIRT_BLOCK / 64);
            // How a batch's path ids reach the waves (profiles/r02_small_launches.txt).  Two things pull apart: a single atomic word
            // hands out only ~90 M claims a second, while the list walk ends 10..25 G paths a second -- so claims must be large, or
            // part of every wave's fair share must be its own from the start (TraceArgs::own_chunks: chunks of 64 ids dealt
            // round-robin over the waves, only the rest is claimed); but waves do not advance at equal speed, so what is dealt
            // statically or claimed in large pieces lengthens the end of the launch (Cornell 1024^2@512 with equal static shares,
            // however dealt and at any grid size: 56 -> 75 ms).  Measured optimum by launch size:
            //   list walk   up to ~1.2 M paths (the capped grid above): claims of 64, six eighths of the share static;
            //               up to 12 M: claims of 256, four eighths static; up to 32 M: 256, three eighths (Cornell 1024^2@16:
            //               2.45 -> 2.12 ms, 4 spheres 1200x800@10: 0.94 -> 0.64 ms); beyond: ~32 claims per wave, 256..512 ids, none static
            //               (what a wave still holds when the cursor runs dry is the tail of the launch: exits spread over 0.72 ms with
            //               claims of 2048, 0.65 with 512; claims of 256 throughout gain 0.7 % on Cornell 1024^2@128 and lose 38 % on the
            //               4-sphere scene at 1200x800@500, whose 29 G paths a second then ask the counter for 115 M claims a second);
            //   BVH walks   (longer paths of more uneven length): up to 6 M paths claims of 64 (40 M in trees of 64 Ki nodes and more,
            //               whose walks are slower still), up to 64 M of 128, three eighths static in both (485 spheres 1200x800@32: 5.55 -> 5.03 ms); beyond: claims of 256, none static.
            uint32_t own_eighths = 0;
            if (mode == 0) {
                if (a.n_paths <= 3 * per_share) { a.claim = 64; own_eighths = 6; }
                else if (a.n_paths <= (32u << 20)) { a.claim = 256; own_eighths = a.n_paths <= (12u << 20) ? 4 : 3; }
                else a.claim = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(256, (a.n_paths / (waves * 32)) & ~63ull));
            } else {
                if (a.n_paths <= (64u << 20)) { a.claim = a.n_paths <= ((a.sc.n_nodes >= 65536u ? 40u : 6u) << 20) ? 64 : 128; own_eighths = 3; }
                else a.claim = 256;
            }
            const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)a.n_paths + a.claim - 1) / a.claim, grid_cap);
            const uint64_t grid_waves = (uint64_t)grid * (MIRT_BLOCK / 64);
            a.grid_waves = (uint32_t)grid_waves;
            a.own_chunks = (uint32_t)(a.n_paths / grid_waves / 64 * own_eighths / 8);
            a.first_claim = (uint32_t)(grid_waves * a.own_chunks * 64);
            if (timed) HIPCHK(c, hipEventRecord(c->ev_pool[2 * batch], st));
            launch_trace(c, mode, a, grid, st);
            if (timed) HIPCHK(c, hipEventRecord(c->ev_pool[2 * batch + 1], st));
            hipLaunchKernelGGL(k_accumulate, dim3((np + MIRT_BLOCK - 1) / MIRT_BLOCK), dim3(MIRT_BLOCK), 0, st,
                               (const RadSlot *)a.rad, d_sum, pix0, np, ns, one_spp ? 1u : 0u, a.tiles_x, a.width);
        }
    }
    HIPCHK(c, hipGetLastError());
    if (stats) {
        HIPCHK(c, hipEventRecord(c->ev1, st));
        HIPCHK(c, hipStreamSynchronize(st));
        float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
#if (MIRT_TEXP & 16)
        if (const char *path = getenv("MIRT_WAVE_DUMP")) {
            std::vector<uint32_t> w((size_t)max_grid * (MIRT_BLOCK / 64) * 8);
            HIPCHK(c, hipMemcpy(w.data(), c->d_mdq.p, w.size() * 4, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "wb")) { fwrite(w.data(), 4, w.size(), f); fclose(f); }
        }
#endif
        std::vector<uint32_t> h(n_batches * per_batch);
        HIPCHK(c, hipMemcpy(h.data(), counts, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        stats->device_ms = ms; stats->bounce_launches = n_batches; stats->batches = n_batches;
        stats->samples = npix_total * spp_count;
        if (timed) for (uint32_t b = 0; b < n_batches; ++b) {          // k_trace launches are reported in the extend_* fields
            float t = 0; HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[2 * b], c->ev_pool[2 * b + 1]));
            stats->extend_ms += t; stats->extend_launches++;
        }
        for (uint32_t b = 0; b < n_batches; ++b) {
            const uint32_t *hist = h.data() + b * per_batch + 1;
            uint64_t ended_at_or_after = 0;                      // paths that traced a ray at depth d = those ending at depth >= d
            for (int d = 63; d >= 0; --d) { ended_at_or_after += hist[d]; stats->depth_hist[d] += ended_at_or_after; }
            stats->depth_hist[63] += hist[64];
        }
        for (int d = 0; d < 64; ++d) stats->rays += stats->depth_hist[d];
    }
    return MIRT_OK;
}

static PathQ carve_queue(float4 *base, size_t cap)
{
    PathQ q;
    q.p0 = base; q.p1 = base + cap; q.p2 = base + 2 * cap;
    return q;
}

// The argument checks mirt_render_device and mirt_render_device_noalloc share.
static int check_render_args(mirt_ctx *c, const mirt_render_params *p)
{
    if (!c->have_scene || !c->have_camera) return fail(c, MIRT_ERR_STATE, "mirt_render_device: upload a scene and set a camera first");
    if (!p->width || !p->height) return fail(c, MIRT_ERR_ARG, "mirt_render_device: empty image");
    if ((uint64_t)p->width * p->height > (1ull << 31)) return fail(c, MIRT_ERR_ARG, "mirt_render_device: image larger than 2^31 pixels");
    if (p->max_depth > 4096) return fail(c, MIRT_ERR_ARG, "mirt_render_device: max_depth > 4096");
    if (!(p->flags & MIRT_FLAG_ONE_SPP) && (uint64_t)p->spp_begin + p->spp_count > 0xFFFFFFFFull) return fail(c, MIRT_ERR_ARG, "mirt_render_device: sample index overflows 32 bits");
    return MIRT_OK;
}

// mirt_render_device for callers that cannot allocate (HIP-graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the same kernels, their scratch carved out of the file-scope `scratch` array of this GPU.  Calls no
// hipMalloc / hipFree / allocating routine.  The array is shared by every call on the GPU and keeps the last call's
// contents: the counters are cleared on `stream` first (radiance slots and stack levels are written before they are
// read), and calls that overlap on one GPU race.
extern "C" int mirt_render_device_noalloc(mirt_ctx *c, const mirt_render_params *p, float *d_sum, void *stream)
{
    if (!c || !p || !d_sum) return MIRT_ERR_ARG;
    CHECK_SIZE(c, p, mirt_render_params);
    int rc = check_render_args(c, p);
    if (rc) return rc;
    if (c->engine != 1) return fail(c, MIRT_ERR_STATE, "mirt_render_device_noalloc: the persistent engine only (option engine=1)");
    if (!c->scratch_dev) return fail(c, MIRT_ERR_STATE, "mirt_render_device_noalloc: no scratch on this device (device id >= 64)");
    const bool one_spp = p->flags & MIRT_FLAG_ONE_SPP;
    const uint32_t spp_count = one_spp ? 1u : p->spp_count;
    const uint32_t spp_begin = one_spp ? 0u : p->spp_begin;
    if (spp_count == 0) return MIRT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    Workspace w;
    w.rad = (RadSlot *)c->scratch_dev; w.rad_paths = SCRATCH_RAD_PATHS;
    w.counts = (uint32_t *)(c->scratch_dev + SCRATCH_RAD_BYTES); w.counts_bytes = SCRATCH_COUNTS_BYTES;
    w.spill = (int32_t *)(c->scratch_dev + SCRATCH_RAD_BYTES + SCRATCH_COUNTS_BYTES); w.spill_bytes = SCRATCH_SPILL_BYTES;
    return render_persistent(c, p, d_sum, (hipStream_t)stream, nullptr, accel_mode(c, p->flags), spp_begin, spp_count, one_spp, &w);
}

extern "C" int mirt_render_device(mirt_ctx *c, const mirt_render_params *p, float *d_sum, void *stream, mirt_stats *stats)
{
    if (!c || !p || !d_sum) return MIRT_ERR_ARG;
    CHECK_SIZE(c, p, mirt_render_params); CHECK_SIZE(c, stats, mirt_stats);
    int rc_args = check_render_args(c, p);
    if (rc_args) return rc_args;
    const uint64_t npix_total = (uint64_t)p->width * p->height;
    const bool one_spp = p->flags & MIRT_FLAG_ONE_SPP;
    const uint32_t spp_count = one_spp ? 1u : p->spp_count;
    const uint32_t spp_begin = one_spp ? 0u : p->spp_begin;
    if (stats) { memset(stats, 0, sizeof *stats); stats->struct_size = (uint32_t)sizeof *stats; }
    if (spp_count == 0) return MIRT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int mode = accel_mode(c, p->flags);
    if (c->engine == 1) return render_persistent(c, p, d_sum, st, stats, mode, spp_begin, spp_count, one_spp);
    if (c->sc.n_textures) return fail(c, MIRT_ERR_STATE, "mirt_render_device: the wavefront engine (option engine=0) renders single-colour scenes only; this scene has image textures");
    if (c->has_chains) return fail(c, MIRT_ERR_STATE, "mirt_render_device: the wavefront engine (option engine=0) renders the factories' shapes only; this scene has primitives in chain form (general instance chains)");

    // batch geometry: npix_b pixels x S samples <= capacity (the queues cost 96 B per path: at most 256 Mi paths here)
    const uint64_t cap = std::min<uint64_t>(std::min<uint64_t>(c->max_paths, 256ull << 20), npix_total * spp_count);
    const uint32_t npix_b = (uint32_t)std::min<uint64_t>(npix_total, cap);
    const uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spp_count, cap / npix_b));
    const size_t capq = (size_t)npix_b * S;
    const uint32_t n_pix_tiles = (uint32_t)((npix_total + npix_b - 1) / npix_b);
    const uint32_t n_s_tiles = (spp_count + S - 1) / S;
    const uint32_t n_batches = n_pix_tiles * n_s_tiles;
    const uint32_t n_depth = p->max_depth + 2;            // per depth d: [0] slots reserved in, [1] paths entering bounce d, [2] work cursor
    const uint32_t per_cu = c->blocks_per_cu ? c->blocks_per_cu : bounce_occupancy(c, mode);
    if (getenv("MIRT_DEBUG")) fprintf(stderr, "[mirt] mode %d, lds %u B, %u workgroups/CU (occupancy query %u)\n", mode, mode == 2 ? c->lds_bytes2 : c->lds_bytes, per_cu, bounce_occupancy(c, mode));
    const uint32_t max_grid = (uint32_t)c->n_cu * per_cu;
    // out-queue chunk size (slots a workgroup reserves with one atomic) and the slot capacity that makes an
    // overflow impossible: a closed chunk holds > chunk-256 paths, every workgroup leaves one open chunk
    const uint32_t grid_max = (uint32_t)std::min<uint64_t>((capq + MIRT_BLOCK - 1) / MIRT_BLOCK, max_grid);
    const uint32_t chunk_shift = capq >= (4u << 20) ? 12u : (capq >= (1u << 20) ? 11u : 10u);
    const uint64_t chunk = 1ull << chunk_shift;
    const uint64_t n_chunks_max = capq / (chunk - 255) + grid_max + 2;
    const uint64_t cap_slots = n_chunks_max * chunk;
    if (cap_slots > 0xFFFF0000ull) return fail(c, MIRT_ERR_ARG, "mirt_render_device: ray queue larger than 2^32 slots (lower max_paths)");

    if (c->d_queues.ensure(2 * 3 * cap_slots * sizeof(float4)) || c->d_rad.ensure(capq * sizeof(RadSlot)) ||
        c->d_counts.ensure(((size_t)n_batches * n_depth * 3 + 1) * sizeof(uint32_t)) || c->d_len.ensure(2 * n_chunks_max * sizeof(uint32_t)))
        return fail(c, MIRT_ERR_NOMEM, "mirt_render_device: hipMalloc of the ray queues failed (lower the max_paths option)");
    PathQ q[2] = { carve_queue((float4 *)c->d_queues.p, cap_slots), carve_queue((float4 *)c->d_queues.p + 3 * cap_slots, cap_slots) };
    uint32_t *qlen[2] = { (uint32_t *)c->d_len.p, (uint32_t *)c->d_len.p + n_chunks_max };
    uint32_t *counts = (uint32_t *)c->d_counts.p;
    const size_t n_counts = (size_t)n_batches * n_depth * 3 + 1;
    uint32_t *overflow = counts + n_counts - 1;
    HIPCHK(c, hipMemsetAsync(counts, 0, n_counts * sizeof(uint32_t), st));
    if (stats) HIPCHK(c, hipEventRecord(c->ev0, st));

    BounceArgs a{};
    a.sc = mode == 2 ? c->sc2 : c->sc; a.cam = c->cam;
    if (bind_spill(c, a.sc, grid_max)) return fail(c, MIRT_ERR_NOMEM, "mirt_render_device: hipMalloc of the traversal-stack spill area failed"); a.rad = (RadSlot *)c->d_rad.p;
    a.width = p->width; a.height = p->height; a.max_depth = p->max_depth; a.flags = p->flags;
    a.by_width = fastdiv_make(a.width);
    a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
    a.chunk_shift = chunk_shift; a.cap_slots = (uint32_t)cap_slots; a.overflow = overflow;
    uint32_t launches = 0, batch = 0;
    const bool timed = stats && c->time_launches;
    std::vector<uint8_t> ev_first;       // per timed launch: its depth (clamped to 63)
    if (timed) {
        size_t need = 2 * (size_t)n_batches * (p->max_depth + 1);
        while (c->ev_pool.size() < need) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->ev_pool.push_back(e); }
    }
    for (uint32_t pt = 0; pt < n_pix_tiles; ++pt) {
        const uint32_t pix0 = pt * npix_b;
        const uint32_t np = (uint32_t)std::min<uint64_t>(npix_b, npix_total - pix0);
        for (uint32_t stile = 0; stile < n_s_tiles; ++stile, ++batch) {
            const uint32_t s0 = spp_begin + stile * S;
            const uint32_t ns = std::min<uint32_t>(S, spp_begin + spp_count - s0);
            const uint32_t n_first = np * ns;
            uint32_t *cnt = counts + (size_t)batch * n_depth * 3;
            a.pix0 = pix0; a.npix = np; a.spp0 = s0; a.n_first = n_first;
            const uint32_t grid = std::min<uint32_t>((n_first + MIRT_BLOCK - 1) / MIRT_BLOCK, max_grid);
            for (uint32_t d = 0; d <= p->max_depth; ++d) {
                a.depth = d;
                a.in = q[d & 1]; a.out = q[(d + 1) & 1];
                a.res_in = cnt + 3 * d; a.cursor = cnt + 3 * d + 2; a.res_out = cnt + 3 * (d + 1); a.alive_out = cnt + 3 * (d + 1) + 1;
                a.len_in = qlen[d & 1]; a.len_out = qlen[(d + 1) & 1];
                if (timed) HIPCHK(c, hipEventRecord(c->ev_pool[2 * ev_first.size()], st));
                if (d == 0) launch_bounce<true>(c, mode, a, grid, st);
                else launch_bounce<false>(c, mode, a, grid, st);
                if (timed) { HIPCHK(c, hipEventRecord(c->ev_pool[2 * ev_first.size() + 1], st)); ev_first.push_back((uint8_t)std::min<uint32_t>(d, 63u)); }
                ++launches;
                if (p->flags & MIRT_FLAG_NORMAL_DISPLAY) break;      // every path ends at its first vertex
            }
            hipLaunchKernelGGL(k_accumulate, dim3((np + MIRT_BLOCK - 1) / MIRT_BLOCK), dim3(MIRT_BLOCK), 0, st,
                               (const RadSlot *)c->d_rad.p, d_sum, pix0, np, ns, one_spp ? 1u : 0u, 0u, p->width);
        }
    }
    HIPCHK(c, hipGetLastError());
    if (stats) {
        HIPCHK(c, hipEventRecord(c->ev1, st));
        HIPCHK(c, hipStreamSynchronize(st));
        float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        std::vector<uint32_t> h(n_counts);
        HIPCHK(c, hipMemcpy(h.data(), counts, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (h[n_counts - 1]) return fail(c, MIRT_ERR_STATE, "mirt_render_device: ray-queue reservation overflowed (internal bound violated)");
        stats->device_ms = ms; stats->bounce_launches = launches; stats->batches = n_batches;
        for (size_t i = 0; i < ev_first.size(); ++i) {
            float t = 0; HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[2 * i], c->ev_pool[2 * i + 1]));
            stats->depth_ms[ev_first[i]] += t;
            if (ev_first[i] == 0) { stats->first_ms += t; stats->first_launches++; } else { stats->extend_ms += t; stats->extend_launches++; }
        }
        stats->samples = npix_total * spp_count;
        stats->rays = stats->samples;                                  // bounce 0 traces one ray per sample
        stats->depth_hist[0] = stats->samples;
        for (uint32_t b = 0; b < n_batches; ++b)
            for (uint32_t d = 1; d <= p->max_depth; ++d) {
                uint64_t v = h[((size_t)b * n_depth + d) * 3 + 1];
                stats->rays += v; stats->depth_hist[d < 63 ? d : 63] += v;
            }
    }
    return MIRT_OK;
}

extern "C" int mirt_resolve_device(mirt_ctx *c, const float *d_sum, uint32_t w, uint32_t h, uint32_t spp_total, uint32_t flags,
                                   float *d_pixels, uint8_t *d_rgba8, void *stream)
{
    if (!c || !d_sum || !w || !h) return MIRT_ERR_ARG;
    if (!(flags & MIRT_FLAG_ONE_SPP) && spp_total == 0) return fail(c, MIRT_ERR_ARG, "mirt_resolve_device: spp_total is 0");
    if (((uintptr_t)d_rgba8) & 3) return fail(c, MIRT_ERR_ARG, "mirt_resolve_device: rgba8 must be 4-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t npix = w * h;
    hipLaunchKernelGGL(k_resolve, dim3((npix + MIRT_BLOCK - 1) / MIRT_BLOCK), dim3(MIRT_BLOCK), 0, (hipStream_t)stream,
                       d_sum, npix, spp_total, flags, d_pixels, (uint32_t *)d_rgba8);
    HIPCHK(c, hipGetLastError());
    return MIRT_OK;
}

extern "C" int mirt_render(mirt_ctx *c, const mirt_render_params *p, float *h_sum, float *h_pixels, uint8_t *h_rgba8, mirt_stats *stats)
{
    if (!c || !p) return MIRT_ERR_ARG;
    CHECK_SIZE(c, p, mirt_render_params); CHECK_SIZE(c, stats, mirt_stats);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)p->width * p->height;
    if (!npix) return fail(c, MIRT_ERR_ARG, "mirt_render: empty image");
    if (c->d_tmp_sum.ensure(npix * 3 * sizeof(float)) || c->d_tmp_pix.ensure(npix * 3 * sizeof(float)) || c->d_tmp_rgba.ensure(npix * 4))
        return fail(c, MIRT_ERR_NOMEM, "mirt_render: hipMalloc failed");
    HIPCHK(c, hipMemset(c->d_tmp_sum.p, 0, npix * 3 * sizeof(float)));
    int rc = mirt_render_device(c, p, (float *)c->d_tmp_sum.p, nullptr, stats);
    if (rc) return rc;
    rc = mirt_resolve_device(c, (const float *)c->d_tmp_sum.p, p->width, p->height, p->spp_count, p->flags,
                             (float *)c->d_tmp_pix.p, (uint8_t *)c->d_tmp_rgba.p, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipDeviceSynchronize());
    if (h_sum) HIPCHK(c, hipMemcpy(h_sum, c->d_tmp_sum.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (h_pixels) HIPCHK(c, hipMemcpy(h_pixels, c->d_tmp_pix.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (h_rgba8) HIPCHK(c, hipMemcpy(h_rgba8, c->d_tmp_rgba.p, npix * 4, hipMemcpyDeviceToHost));
    return MIRT_OK;
}

// Progressive form of mirt_render for the interactive caller (the viewer loop of RayTracer.cpp:251-254 re-invokes
// TraceRay; here successive calls refine one image): the sum stays on the device between calls.
extern "C" int mirt_render_progressive(mirt_ctx *c, const mirt_render_params *p, float *h_sum, float *h_pixels, uint8_t *h_rgba8, mirt_stats *stats)
{
    if (!c || !p) return MIRT_ERR_ARG;
    CHECK_SIZE(c, p, mirt_render_params); CHECK_SIZE(c, stats, mirt_stats);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)p->width * p->height;
    if (!npix) return fail(c, MIRT_ERR_ARG, "mirt_render_progressive: empty image");
    if (p->flags & MIRT_FLAG_ONE_SPP) return fail(c, MIRT_ERR_ARG, "mirt_render_progressive: MIRT_FLAG_ONE_SPP has nothing to accumulate (use mirt_render)");
    if (p->spp_begin == 0) {
        if (c->d_prog_sum.ensure(npix * 3 * sizeof(float))) return fail(c, MIRT_ERR_NOMEM, "mirt_render_progressive: hipMalloc failed");
        HIPCHK(c, hipMemset(c->d_prog_sum.p, 0, npix * 3 * sizeof(float)));
        c->prog_w = p->width; c->prog_h = p->height; c->prog_spp = 0;
    } else if (p->width != c->prog_w || p->height != c->prog_h || p->spp_begin != c->prog_spp) {
        return fail(c, MIRT_ERR_STATE, "mirt_render_progressive: spp_begin must continue the previous call on the same image size (or be 0 to restart)");
    }
    if ((uint64_t)p->spp_begin + p->spp_count > 0xFFFFFFFFull) return fail(c, MIRT_ERR_ARG, "mirt_render_progressive: sample index overflows 32 bits");
    if (c->d_tmp_pix.ensure(npix * 3 * sizeof(float)) || c->d_tmp_rgba.ensure(npix * 4)) return fail(c, MIRT_ERR_NOMEM, "mirt_render_progressive: hipMalloc failed");
    int rc = mirt_render_device(c, p, (float *)c->d_prog_sum.p, nullptr, stats);
    if (rc) return rc;
    c->prog_spp = p->spp_begin + p->spp_count;
    if (c->prog_spp == 0) return MIRT_OK;
    rc = mirt_resolve_device(c, (const float *)c->d_prog_sum.p, p->width, p->height, c->prog_spp, p->flags,
                             (float *)c->d_tmp_pix.p, (uint8_t *)c->d_tmp_rgba.p, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipDeviceSynchronize());
    if (h_sum) HIPCHK(c, hipMemcpy(h_sum, c->d_prog_sum.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (h_pixels) HIPCHK(c, hipMemcpy(h_pixels, c->d_tmp_pix.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (h_rgba8) HIPCHK(c, hipMemcpy(h_rgba8, c->d_tmp_rgba.p, npix * 4, hipMemcpyDeviceToHost));
    return MIRT_OK;
}

// ------------------------------------------------------------------------------------------ function-level entry points
namespace {
struct Tmp {   // RAII device scratch for the batch hooks
    std::vector<void *> v;
    ~Tmp() { for (void *p : v) (void)hipFree(p); }
    template <class T> T *up(const T *h, size_t n) { void *d = nullptr; if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr; v.push_back(d); if (h && n) (void)hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice); return (T *)d; }
};
}
#define NEED(ptr) if (!(ptr)) return fail(c, MIRT_ERR_NOMEM, "hipMalloc failed in a batch entry point")
#define CHECK_N(n) if ((n) > 0x7FFFFFFFull) return fail(c, MIRT_ERR_ARG, "batch too large")

extern "C" int mirt_aabb_hit_batch(mirt_ctx *c, const float *bmin, const float *bmax, size_t nbox, const float *org, const float *dir,
                                   size_t nray, float t_min, float t_max, uint8_t *out)
{
    if (!c || !bmin || !bmax || !org || !dir || !out) return MIRT_ERR_ARG;
    if (!nray || !nbox) return MIRT_OK;
    CHECK_N(nray); CHECK_N(nbox);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    float *dmin = t.up(bmin, 3 * nbox), *dmax = t.up(bmax, 3 * nbox), *dorg = t.up(org, 3 * nray), *ddir = t.up(dir, 3 * nray);
    uint8_t *dout = t.up<uint8_t>(nullptr, nray * nbox);
    NEED(dmin && dmax && dorg && ddir && dout);
    hipLaunchKernelGGL(k_aabb_batch, dim3((uint32_t)((nray + 255) / 256)), dim3(256), 0, 0, dmin, dmax, (uint32_t)nbox, dorg, ddir, (uint32_t)nray, t_min, t_max, dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out, dout, nray * nbox, hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_prim_hit_batch(mirt_ctx *c, uint32_t prim, const float *org, const float *dir, size_t nray, float t_min, float t_max, mirt_hit *out)
{
    if (!c || !org || !dir || !out) return MIRT_ERR_ARG;
    if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_prim_hit_batch: no scene");
    if (prim >= c->sc.n_prims) return fail(c, MIRT_ERR_ARG, "mirt_prim_hit_batch: primitive out of range");
    if (!nray) return MIRT_OK;
    CHECK_N(nray);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    float *dorg = t.up(org, 3 * nray), *ddir = t.up(dir, 3 * nray);
    mirt_hit *dout = t.up<mirt_hit>(nullptr, nray);
    NEED(dorg && ddir && dout);
    hipLaunchKernelGGL(k_prim_hit_batch, dim3((uint32_t)((nray + 255) / 256)), dim3(256), 0, 0, c->sc, prim, dorg, ddir, (uint32_t)nray, t_min, t_max, dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out, dout, nray * sizeof(mirt_hit), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_world_hit_batch(mirt_ctx *c, const float *org, const float *dir, size_t nray, float t_min, float t_max, uint32_t flags, mirt_hit *out)
{
    if (!c || !org || !dir || !out) return MIRT_ERR_ARG;
    if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_world_hit_batch: no scene");
    if (!nray) return MIRT_OK;
    CHECK_N(nray);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    float *dorg = t.up(org, 3 * nray), *ddir = t.up(dir, 3 * nray);
    mirt_hit *dout = t.up<mirt_hit>(nullptr, nray);
    NEED(dorg && ddir && dout);
    dim3 g((uint32_t)std::min<size_t>((nray + MIRT_BLOCK - 1) / MIRT_BLOCK, (size_t)c->n_cu * 16)), b(MIRT_BLOCK);
    const int mode = accel_mode(c, flags);
    // the list and near-first walks assume t_min > 0 (the sphere "behind the origin" cull, the next-float-up of slab_hit,
    // the estimated-root band); the render path always passes 0.001.  The reference-order walk (no accel flag) takes any t_min.
    if (mode != 1 && !(t_min > 0.0f)) return fail(c, MIRT_ERR_ARG, "mirt_world_hit_batch: t_min must be > 0 with MIRT_FLAG_ACCEL_LIST/_AUTO/_ORDERED (pass no accel flag for t_min <= 0)");
    const bool pl = c->prims_in_lds;
    const uint32_t lds = mode == 2 ? c->lds_bytes2 : (mode == 0 ? c->lds_bytes - c->lds_stack_bytes : c->lds_bytes);
    SceneDev sd = mode == 2 ? c->sc2 : c->sc;
    if (bind_spill(c, sd, g.x)) return fail(c, MIRT_ERR_NOMEM, "mirt_world_hit_batch: hipMalloc of the traversal-stack spill area failed");
    if (mode == 0) { if (pl) hipLaunchKernelGGL((k_world_hit_batch<0, true>), g, b, lds, 0, sd, dorg, ddir, (uint32_t)nray, t_min, t_max, dout); else hipLaunchKernelGGL((k_world_hit_batch<0, false>), g, b, lds, 0, sd, dorg, ddir, (uint32_t)nray, t_min, t_max, dout); }
    else if (mode == 1) { if (pl) hipLaunchKernelGGL((k_world_hit_batch<1, true>), g, b, lds, 0, sd, dorg, ddir, (uint32_t)nray, t_min, t_max, dout); else hipLaunchKernelGGL((k_world_hit_batch<1, false>), g, b, lds, 0, sd, dorg, ddir, (uint32_t)nray, t_min, t_max, dout); }
    else { if (pl) hipLaunchKernelGGL((k_world_hit_batch<2, true>), g, b, lds, 0, sd, dorg, ddir, (uint32_t)nray, t_min, t_max, dout); else hipLaunchKernelGGL((k_world_hit_batch<2, false>), g, b, lds, 0, sd, dorg, ddir, (uint32_t)nray, t_min, t_max, dout); }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out, dout, nray * sizeof(mirt_hit), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_scatter_batch(mirt_ctx *c, const mirt_hit *hits, const float *dir_in, size_t n, const uint32_t *draws, uint32_t dpr,
                                  uint8_t *out_alive, float *out_atten, float *out_org, float *out_dir, float *out_emit, uint32_t *consumed)
{
    if (!c || !hits || !dir_in || !draws || !out_alive || !out_atten || !out_org || !out_dir || !out_emit) return MIRT_ERR_ARG;
    if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_scatter_batch: no scene");
    if (!n) return MIRT_OK;
    CHECK_N(n);
    for (size_t i = 0; i < n; ++i) if (hits[i].mat < 0 || (uint32_t)hits[i].mat >= c->sc.n_mats) return fail(c, MIRT_ERR_ARG, "mirt_scatter_batch: hit with bad material index");
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    mirt_hit *dh = t.up(hits, n); float *dd = t.up(dir_in, 3 * n); uint32_t *dr = t.up(draws, n * (size_t)dpr);
    uint8_t *da = t.up<uint8_t>(nullptr, n); float *dat = t.up<float>(nullptr, 3 * n), *dorg = t.up<float>(nullptr, 3 * n), *ddir = t.up<float>(nullptr, 3 * n), *dem = t.up<float>(nullptr, 3 * n);
    uint32_t *dc = t.up<uint32_t>(nullptr, n);
    NEED(dh && dd && dr && da && dat && dorg && ddir && dem && dc);
    hipLaunchKernelGGL(k_scatter_batch, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, c->sc, dh, dd, (uint32_t)n, dr, dpr, da, dat, dorg, ddir, dem, dc);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out_alive, da, n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_atten, dat, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_org, dorg, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_dir, ddir, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_emit, dem, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (consumed) HIPCHK(c, hipMemcpy(consumed, dc, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_camera_rays(mirt_ctx *c, const float *uv, size_t n, const uint32_t *draws, uint32_t dpr, float *out_org, float *out_dir, uint32_t *consumed)
{
    if (!c || !uv || !draws || !out_org || !out_dir) return MIRT_ERR_ARG;
    if (!c->have_camera) return fail(c, MIRT_ERR_STATE, "mirt_camera_rays: no camera");
    if (!n) return MIRT_OK;
    CHECK_N(n);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    float *duv = t.up(uv, 2 * n); uint32_t *dr = t.up(draws, n * (size_t)dpr);
    float *dorg = t.up<float>(nullptr, 3 * n), *ddir = t.up<float>(nullptr, 3 * n); uint32_t *dc = t.up<uint32_t>(nullptr, n);
    NEED(duv && dr && dorg && ddir && dc);
    hipLaunchKernelGGL(k_camera_rays, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, c->cam, duv, (uint32_t)n, dr, dpr, dorg, ddir, dc);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out_org, dorg, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_dir, ddir, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (consumed) HIPCHK(c, hipMemcpy(consumed, dc, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_philox_batch(mirt_ctx *c, const uint32_t *ctr, size_t n, const uint32_t key[2], uint32_t *out)
{
    if (!c || !ctr || !key || !out) return MIRT_ERR_ARG;
    if (!n) return MIRT_OK;
    CHECK_N(n);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    uint32_t *dc = t.up(ctr, 4 * n), *dout = t.up<uint32_t>(nullptr, 4 * n);
    NEED(dc && dout);
    hipLaunchKernelGGL(k_philox_batch, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, dc, (uint32_t)n, key[0], key[1], dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out, dout, 4 * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_tex_sample_batch(mirt_ctx *c, uint32_t tex, const float *uv, size_t n, float *rgb_out)
{
    if (!c || !uv || !rgb_out) return MIRT_ERR_ARG;
    if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_tex_sample_batch: no scene");
    if (tex >= c->sc.n_textures) return fail(c, MIRT_ERR_ARG, "mirt_tex_sample_batch: texture index out of range");
    if (!n) return MIRT_OK;
    CHECK_N(n);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    float *duv = t.up(uv, 2 * n), *dout = t.up<float>(nullptr, 3 * n);
    NEED(duv && dout);
    hipLaunchKernelGGL(k_tex_sample, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, c->sc, tex, duv, (uint32_t)n, dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(rgb_out, dout, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_arith_batch(mirt_ctx *c, uint32_t op, const float *x, const float *y, size_t n, float *out)
{
    if (!c || !x || !out || op > 9 || ((op == 0 || op == 5 || op == 6 || op == 9) && !y)) return MIRT_ERR_ARG;
    if (!n) return MIRT_OK;
    CHECK_N(n);
    HIPCHK(c, hipSetDevice(c->device));
    Tmp t;
    float *dx = t.up(x, n), *dy = y ? t.up(y, n) : nullptr, *dout = t.up<float>(nullptr, n);
    NEED(dx && dout && (!y || dy));
    hipLaunchKernelGGL(k_arith, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, op, dx, dy, (uint32_t)n, dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

// ------------------------------------------------------------------------------------------ PPMImageMaker.cpp:9-48
extern "C" int mirt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgba8)
{
    if (!path || !rgba8) return MIRT_ERR_ARG;
    FILE *f = fopen(path, "w");
    if (!f) return MIRT_ERR_ARG;
    std::string buf;
    buf.reserve(1 << 20);
    char line[64];
    int n = snprintf(line, sizeof line, "P3\n%u %u\n255\n", width, height);
    buf.append(line, n);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        n = snprintf(line, sizeof line, "%u %u %u\n", (unsigned)rgba8[4 * i], (unsigned)rgba8[4 * i + 1], (unsigned)rgba8[4 * i + 2]);
        buf.append(line, n);
        if (buf.size() > (1 << 20) - 64) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    }
    fwrite(buf.data(), 1, buf.size(), f);
    return fclose(f) == 0 ? MIRT_OK : MIRT_ERR_ARG;
}

extern "C" int mirt_write_image(const char *path, int format, uint32_t width, uint32_t height, const uint8_t *rgba8, const float *rgb_float)
{
    if (!path || !width || !height) return MIRT_ERR_ARG;
    if (format == MIRT_IMG_P3) return mirt_write_ppm(path, width, height, rgba8);
    if (format != MIRT_IMG_P6 && format != MIRT_IMG_PFM) return MIRT_ERR_ARG;
    if (format == MIRT_IMG_P6 ? !rgba8 : !rgb_float) return MIRT_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return MIRT_ERR_ARG;
    bool ok = true;
    if (format == MIRT_IMG_P6) {
        ok = fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;
        std::vector<uint8_t> row(3 * (size_t)width);
        for (uint32_t y = 0; y < height && ok; ++y) {
            const uint8_t *src = rgba8 + 4 * (size_t)y * width;
            for (uint32_t x = 0; x < width; ++x) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
            ok = fwrite(row.data(), 1, row.size(), f) == row.size();
        }
    } else {
        const uint16_t probe = 1;
        if (*(const uint8_t *)&probe != 1) { fclose(f); return MIRT_ERR_ARG; }          // little-endian hosts only
        ok = fprintf(f, "PF\n%u %u\n-1.0\n", width, height) > 0;
        for (uint32_t y = height; y-- > 0 && ok;)
            ok = fwrite(rgb_float + 3 * (size_t)y * width, sizeof(float), 3 * (size_t)width, f) == 3 * (size_t)width;
    }
    return (fclose(f) == 0 && ok) ? MIRT_OK : MIRT_ERR_ARG;
}
