// mirt_multi.hip -- one render sharded over the GPUs of a node, inside one process (include/mirt.h, "multi-GPU").
//
// The reference parallelises TraceRay over image rows with OpenMP and no communication
// (HomemadeRayTracer.cpp:107-110).  Here the unit that shards is the sample: device g of G renders the global sample
// indices [g*spp/G, (g+1)*spp/G) of EVERY pixel (the Philox counter holds the global index, so the set of samples
// does not depend on G), one host thread per device drives its own mirt_ctx, and the G float radiance-sum buffers
// are summed onto device 0 -- the only exchange of the path -- which then resolves the image.
//
// Two ways to do that sum:
//   RCCL   one ncclReduce(sum, f32, root 0) per device in a group (ring/tree over xGMI, RCCL's choice); needs
//          distinct devices; librccl.so is opened with dlopen so that a single-GPU host does not need it.
//   peer   device 0 pulls each peer's buffer with hipMemcpyPeerAsync (each peer has its own xGMI link to device 0,
//          so the G-1 copies run side by side) and adds them in rank order: a fixed association, hence an image that
//          is bit-reproducible for a given G; works with a device listed twice (how the tests run it on one GPU).
// Either way the image differs from the one-device image only by the re-association of the per-pixel sum.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
// librccl is opened with dlopen, so a host without it (or without its headers) still builds and runs the single-GPU product and
// the peer reduce: the header is used when it is there, else the handful of declarations the four entry points need (values
// as in rccl.h / nccl.h, checked against the header by the static_asserts below whenever it is present)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(ncclSuccess == 0 && ncclFloat32 == 7 && ncclSum == 0, "the local declarations below must follow rccl.h");
#else
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif
#include <chrono>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mirt.h"

namespace {

thread_local std::string g_multi_error;

__global__ void __launch_bounds__(256) k_add_into(float *__restrict__ dst, const float *__restrict__ src, size_t n4, size_t n)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { f32x4 a = ((const f32x4 *)dst)[i], b = __builtin_nontemporal_load((const f32x4 *)src + i); ((f32x4 *)dst)[i] = a + b; }
    if (i == 0) for (size_t k = n4 * 4; k < n; ++k) dst[k] += src[k];      // tail of a buffer whose length is not a multiple of 4
}

struct Rccl {                                                              // the four entry points used, resolved at run time
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::mutex mu;                                                         // contexts of different host threads may arrive here together
    bool load()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (so) return true;
        so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!so) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(so, "ncclReduce");
        GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
        if (CommInitAll && CommDestroy && Reduce && GroupStart && GroupEnd && GetErrorString) return true;
        dlclose(so); so = nullptr;
        return false;
    }
};
Rccl g_rccl;

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

struct mirt_multi {
    std::vector<int> dev;
    std::vector<mirt_ctx *> ctx;
    std::vector<hipStream_t> stream;          // per rank, on its device
    std::vector<float *> d_sum;               // per rank, width*height*3 floats
    std::vector<float *> d_stage;             // peer reduce: on device 0, one per peer
    std::vector<hipStream_t> copy_stream;     // peer reduce: on device 0, one per peer
    std::vector<hipEvent_t> copy_done;
    std::vector<ncclComm_t> comm;             // RCCL reduce
    float *d_pixels = nullptr; uint8_t *d_rgba = nullptr;
    size_t n_floats = 0;                      // capacity of the buffers above
    int reduce_want = MIRT_REDUCE_AUTO, reduce_used = 0;
    bool auto_peer = false;                   // MIRT_REDUCE_AUTO found that RCCL cannot set up its communicators here: the peer reduce from then on
    bool distinct = true;
    mirt_multi_info last{};
    std::string err;
};

#define MCHK(m, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) { (m)->err = std::string(#call) + ": " + hipGetErrorString(e_); return MIRT_ERR_HIP; } \
    } while (0)

static int mfail(mirt_multi *m, int code, const std::string &msg) { m->err = msg; return code; }

extern "C" const char *mirt_multi_last_error(const mirt_multi *m) { return m ? m->err.c_str() : g_multi_error.c_str(); }

static void release_buffers(mirt_multi *m)
{
    for (size_t i = 0; i < m->dev.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        if (i < m->d_sum.size() && m->d_sum[i]) (void)hipFree(m->d_sum[i]);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    for (float *p : m->d_stage) if (p) (void)hipFree(p);
    if (m->d_pixels) (void)hipFree(m->d_pixels);
    if (m->d_rgba) (void)hipFree(m->d_rgba);
    m->d_sum.assign(m->dev.size(), nullptr); m->d_stage.assign(m->dev.size(), nullptr);
    m->d_pixels = nullptr; m->d_rgba = nullptr; m->n_floats = 0;
}

extern "C" void mirt_multi_destroy(mirt_multi *m)
{
    if (!m) return;
    release_buffers(m);
    for (ncclComm_t c : m->comm) if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < m->dev.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        if (i < m->stream.size() && m->stream[i]) (void)hipStreamDestroy(m->stream[i]);
        if (i < m->ctx.size() && m->ctx[i]) mirt_destroy(m->ctx[i]);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    for (hipStream_t s : m->copy_stream) if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : m->copy_done) if (e) (void)hipEventDestroy(e);
    delete m;
}

extern "C" int mirt_multi_create(const int *device_ids, int n, mirt_multi **out)
{
    if (!out) { g_multi_error = "mirt_multi_create: out is NULL"; return MIRT_ERR_ARG; }
    *out = nullptr;
    if (!device_ids || n < 1 || n > MIRT_MULTI_MAX_DEVICES) { g_multi_error = "mirt_multi_create: need 1.." + std::to_string(MIRT_MULTI_MAX_DEVICES) + " device ids"; return MIRT_ERR_ARG; }
    mirt_multi *m = new mirt_multi();
    m->dev.assign(device_ids, device_ids + n);
    m->distinct = std::set<int>(m->dev.begin(), m->dev.end()).size() == (size_t)n;
    m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr); m->d_sum.assign(n, nullptr);
    m->d_stage.assign(n, nullptr); m->copy_stream.assign(n, nullptr); m->copy_done.assign(n, nullptr);
    for (int i = 0; i < n; ++i) {
        int rc = mirt_create(m->dev[i], &m->ctx[i]);                      // makes device i current
        if (rc != MIRT_OK) { g_multi_error = std::string("mirt_multi_create: device ") + std::to_string(m->dev[i]) + ": " + mirt_last_error(nullptr); mirt_multi_destroy(m); return rc; }
        if (hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking) != hipSuccess) { g_multi_error = "mirt_multi_create: hipStreamCreate failed"; mirt_multi_destroy(m); return MIRT_ERR_HIP; }
    }
    (void)hipSetDevice(m->dev[0]);
    for (int i = 1; i < n; ++i) {
        if (hipStreamCreateWithFlags(&m->copy_stream[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->copy_done[i], hipEventDisableTiming) != hipSuccess) {
            g_multi_error = "mirt_multi_create: hipStreamCreate / hipEventCreate failed"; mirt_multi_destroy(m); return MIRT_ERR_HIP;
        }
        if (m->dev[i] != m->dev[0]) {                                      // direct xGMI copies device i -> device 0
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->dev[0], m->dev[i]) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(m->dev[i], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // hipMemcpyPeer then stages through the host
                else (void)hipGetLastError();
            }
        }
    }
    *out = m;
    return MIRT_OK;
}

extern "C" int mirt_multi_device_count(const mirt_multi *m) { return m ? (int)m->dev.size() : 0; }

extern "C" int mirt_multi_set_option(mirt_multi *m, const char *name, int64_t value)
{
    if (!m || !name) return MIRT_ERR_ARG;
    if (!strcmp(name, "reduce")) {
        if (value != MIRT_REDUCE_AUTO && value != MIRT_REDUCE_PEER && value != MIRT_REDUCE_RCCL) return mfail(m, MIRT_ERR_ARG, "reduce must be MIRT_REDUCE_AUTO, _PEER or _RCCL");
        m->reduce_want = (int)value;
        return MIRT_OK;
    }
    for (mirt_ctx *c : m->ctx) {
        int rc = mirt_set_option(c, name, value);
        if (rc != MIRT_OK) return mfail(m, rc, mirt_last_error(c));
    }
    return MIRT_OK;
}

extern "C" int mirt_multi_upload_scene(mirt_multi *m, const mirt_scene *scene)
{
    if (!m || !scene) return MIRT_ERR_ARG;
    for (mirt_ctx *c : m->ctx) {                                           // the scene (KB..MB) is replicated on every device
        int rc = mirt_upload_scene(c, scene);
        if (rc != MIRT_OK) return mfail(m, rc, mirt_last_error(c));
    }
    return MIRT_OK;
}

extern "C" int mirt_multi_set_camera(mirt_multi *m, const mirt_camera *cam)
{
    if (!m || !cam) return MIRT_ERR_ARG;
    for (mirt_ctx *c : m->ctx) {
        int rc = mirt_set_camera(c, cam);
        if (rc != MIRT_OK) return mfail(m, rc, mirt_last_error(c));
    }
    return MIRT_OK;
}

extern "C" void mirt_multi_spp_range(uint32_t rank, uint32_t world, uint32_t spp_begin, uint32_t spp_count, uint32_t *begin, uint32_t *count)
{
    const uint64_t b = (uint64_t)spp_count * rank / world, e = (uint64_t)spp_count * (rank + 1) / world;   // sharding.py spp_range
    if (begin) *begin = spp_begin + (uint32_t)b;
    if (count) *count = (uint32_t)(e - b);
}

static int ensure_buffers(mirt_multi *m, size_t n_floats, size_t npix)
{
    if (n_floats <= m->n_floats) return MIRT_OK;
    release_buffers(m);
    const size_t n = m->dev.size();
    for (size_t i = 0; i < n; ++i) {
        MCHK(m, hipSetDevice(m->dev[i]));
        MCHK(m, hipMalloc((void **)&m->d_sum[i], n_floats * sizeof(float)));
    }
    MCHK(m, hipSetDevice(m->dev[0]));
    MCHK(m, hipMalloc((void **)&m->d_pixels, n_floats * sizeof(float)));
    MCHK(m, hipMalloc((void **)&m->d_rgba, npix * 4));
    m->n_floats = n_floats;
    return MIRT_OK;
}

static int reduce_peer(mirt_multi *m, size_t n_floats, const std::vector<bool> &has)
{
    MCHK(m, hipSetDevice(m->dev[0]));
    const size_t n = m->dev.size(), bytes = n_floats * sizeof(float);
    for (size_t i = 1; i < n; ++i)                                         // staging buffers on device 0, only when this path runs (RCCL needs none)
        if (has[i] && !m->d_stage[i]) MCHK(m, hipMalloc((void **)&m->d_stage[i], m->n_floats * sizeof(float)));
    for (size_t i = 1; i < n; ++i) {
        if (!has[i]) continue;
        MCHK(m, hipMemcpyPeerAsync(m->d_stage[i], m->dev[0], m->d_sum[i], m->dev[i], bytes, m->copy_stream[i]));
        MCHK(m, hipEventRecord(m->copy_done[i], m->copy_stream[i]));
    }
    const size_t n4 = n_floats / 4;
    for (size_t i = 1; i < n; ++i) {                                       // rank order: ((s0 + s1) + s2) + ...
        if (!has[i]) continue;
        MCHK(m, hipStreamWaitEvent(m->stream[0], m->copy_done[i], 0));
        hipLaunchKernelGGL(k_add_into, dim3((uint32_t)((std::max<size_t>(n4, 1) + 255) / 256)), dim3(256), 0, m->stream[0], m->d_sum[0], m->d_stage[i], n4, n_floats);
    }
    MCHK(m, hipGetLastError());
    MCHK(m, hipStreamSynchronize(m->stream[0]));
    return MIRT_OK;
}

// the communicators, once per mirt_multi; nothing has touched the sums when this fails
static int rccl_init(mirt_multi *m)
{
    const size_t n = m->dev.size();
    if (!m->comm.empty()) return MIRT_OK;
    m->comm.assign(n, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), (int)n, m->dev.data());
    if (r != ncclSuccess) { m->comm.clear(); return mfail(m, MIRT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
    return MIRT_OK;
}
static int reduce_rccl(mirt_multi *m, size_t n_floats)
{
    const size_t n = m->dev.size();
    ncclResult_t r = g_rccl.GroupStart();
    for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
        (void)hipSetDevice(m->dev[i]);
        r = g_rccl.Reduce(m->d_sum[i], m->d_sum[i], n_floats, ncclFloat32, ncclSum, 0, m->comm[i], m->stream[i]);   // in place on the root
    }
    ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return mfail(m, MIRT_ERR_HIP, std::string("ncclReduce: ") + g_rccl.GetErrorString(r));
    for (size_t i = 0; i < n; ++i) { MCHK(m, hipSetDevice(m->dev[i])); MCHK(m, hipStreamSynchronize(m->stream[i])); }
    return MIRT_OK;
}

extern "C" int mirt_multi_render(mirt_multi *m, const mirt_render_params *p, float *h_sum, float *h_pixels, uint8_t *h_rgba8, mirt_stats *stats, mirt_multi_info *info)
{
    if (!m || !p) return MIRT_ERR_ARG;
    if (p->struct_size != sizeof(mirt_render_params)) return mfail(m, MIRT_ERR_ARG, "mirt_multi_render: mirt_render_params.struct_size does not match this library");
    if (stats && stats->struct_size != sizeof(mirt_stats)) return mfail(m, MIRT_ERR_ARG, "mirt_multi_render: mirt_stats.struct_size does not match this library");
    if (info && info->struct_size != sizeof(mirt_multi_info)) return mfail(m, MIRT_ERR_ARG, "mirt_multi_render: mirt_multi_info.struct_size does not match this library");
    if (!p->width || !p->height) return mfail(m, MIRT_ERR_ARG, "mirt_multi_render: empty image");
    const size_t npix = (size_t)p->width * p->height, n_floats = npix * 3, n = m->dev.size();
    int rc = ensure_buffers(m, n_floats, npix);
    if (rc) return rc;

    // 1. every device renders its sample range into its own zeroed sum (one host thread per device)
    const bool one_spp = p->flags & MIRT_FLAG_ONE_SPP;
    std::vector<mirt_render_params> rp(n, *p);
    std::vector<mirt_stats> rs(n);
    std::vector<int> rrc(n, MIRT_OK);
    std::vector<bool> has(n, false);
    for (size_t i = 0; i < n; ++i) {
        mirt_multi_spp_range((uint32_t)i, (uint32_t)n, p->spp_begin, p->spp_count, &rp[i].spp_begin, &rp[i].spp_count);
        has[i] = one_spp ? i == 0 : rp[i].spp_count != 0;                  // 1-SPP mode is one un-jittered sample: device 0 alone
        memset(&rs[i], 0, sizeof rs[i]); rs[i].struct_size = (uint32_t)sizeof rs[i];
    }
    const double t0 = now_ms();
    auto work = [&](size_t i) {
        if (hipSetDevice(m->dev[i]) != hipSuccess || hipMemsetAsync(m->d_sum[i], 0, n_floats * sizeof(float), m->stream[i]) != hipSuccess) { rrc[i] = MIRT_ERR_HIP; return; }
        if (has[i]) rrc[i] = mirt_render_device(m->ctx[i], &rp[i], m->d_sum[i], m->stream[i], &rs[i]);   // synchronises its stream (stats != NULL)
        else if (hipStreamSynchronize(m->stream[i]) != hipSuccess) rrc[i] = MIRT_ERR_HIP;
    };
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < n; ++i) th.emplace_back(work, i);
        work(0);
        for (std::thread &t : th) t.join();
    }
    for (size_t i = 0; i < n; ++i) if (rrc[i] != MIRT_OK) return mfail(m, rrc[i], std::string("device ") + std::to_string(m->dev[i]) + ": " + (rrc[i] == MIRT_ERR_HIP && !*mirt_last_error(m->ctx[i]) ? "HIP call failed" : mirt_last_error(m->ctx[i])));
    const double t1 = now_ms();

    // 2. the one exchange: sum onto device 0
    int used = MIRT_REDUCE_PEER;
    if (n > 1 || m->reduce_want == MIRT_REDUCE_RCCL) {                     // RCCL asked for by name also runs with one device (a one-rank reduce)
        const bool want_rccl = m->reduce_want == MIRT_REDUCE_RCCL || (m->reduce_want == MIRT_REDUCE_AUTO && m->distinct);
        if (want_rccl && !m->distinct) return mfail(m, MIRT_ERR_ARG, "mirt_multi_render: RCCL needs distinct devices (a device is listed twice); use MIRT_REDUCE_PEER");
        if (want_rccl && !m->auto_peer && g_rccl.load()) used = MIRT_REDUCE_RCCL;
        else if (m->reduce_want == MIRT_REDUCE_RCCL) { const char *why = dlerror(); return mfail(m, MIRT_ERR_STATE, std::string("mirt_multi_render: cannot load librccl.so: ") + (why ? why : "missing symbols")); }
        if (used == MIRT_REDUCE_RCCL && (rc = rccl_init(m)) != MIRT_OK) {
            // AUTO promised "RCCL ..., else PEER": a library that loads but cannot build its communicators here (no usable
            // transport between these devices, say) is the "else"; remembered, so later renders do not try again
            if (m->reduce_want != MIRT_REDUCE_AUTO) return rc;
            m->auto_peer = true; used = MIRT_REDUCE_PEER; m->err.clear();
        }
        rc = used == MIRT_REDUCE_RCCL ? reduce_rccl(m, n_floats) : reduce_peer(m, n_floats, has);
        if (rc) return rc;
    }
    const double t2 = now_ms();

    // 3. device 0 resolves (mean over all samples, gamma, RGBA8) and hands the image over
    MCHK(m, hipSetDevice(m->dev[0]));
    rc = mirt_resolve_device(m->ctx[0], m->d_sum[0], p->width, p->height, p->spp_count, p->flags, m->d_pixels, m->d_rgba, m->stream[0]);
    if (rc) return mfail(m, rc, mirt_last_error(m->ctx[0]));
    MCHK(m, hipStreamSynchronize(m->stream[0]));
    if (h_sum) MCHK(m, hipMemcpy(h_sum, m->d_sum[0], n_floats * sizeof(float), hipMemcpyDeviceToHost));
    if (h_pixels) MCHK(m, hipMemcpy(h_pixels, m->d_pixels, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    if (h_rgba8) MCHK(m, hipMemcpy(h_rgba8, m->d_rgba, npix * 4, hipMemcpyDeviceToHost));
    const double t3 = now_ms();

    mirt_multi_info &L = m->last;
    memset(&L, 0, sizeof L); L.struct_size = (uint32_t)sizeof L;
    L.n_devices = (uint32_t)n; L.reduce_kind = (n > 1 || m->reduce_want == MIRT_REDUCE_RCCL) ? (uint32_t)used : 0u;
    L.render_ms = t1 - t0; L.reduce_ms = t2 - t1; L.resolve_ms = t3 - t2; L.reduce_bytes = n > 1 ? (uint64_t)n_floats * sizeof(float) : 0;
    for (size_t i = 0; i < n; ++i) { L.spp_begin[i] = rp[i].spp_begin; L.spp_count[i] = has[i] ? (one_spp ? 1u : rp[i].spp_count) : 0u; L.device_ms[i] = rs[i].device_ms; L.device[i] = m->dev[i]; }
    if (info) *info = L;
    if (stats) {
        memset(stats, 0, sizeof *stats); stats->struct_size = (uint32_t)sizeof *stats;
        for (size_t i = 0; i < n; ++i) {
            stats->rays += rs[i].rays; stats->samples += rs[i].samples;
            for (int d = 0; d < 64; ++d) stats->depth_hist[d] += rs[i].depth_hist[d];
            stats->device_ms = std::max(stats->device_ms, rs[i].device_ms);
            stats->extend_ms = std::max(stats->extend_ms, rs[i].extend_ms);
            stats->bounce_launches += rs[i].bounce_launches; stats->batches += rs[i].batches; stats->extend_launches += rs[i].extend_launches;
        }
    }
    return MIRT_OK;
}
