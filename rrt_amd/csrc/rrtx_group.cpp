// One frame over several devices of one node (include/rrtx.h, "rrtx_group"): N contexts driven by one host
// thread, row-tile shards, ONE grouped RCCL exchange to the first device, a de-interleave pass there.
//
// Nothing in the reference corresponds to this (it selects a single device, main.cpp:107-110); the shape is
// BASELINE.json's: "row-tile-partitioned across the 8 GPUs of one node with a final RCCL gather over xGMI".
// Why send / recv and not ncclAllGather: only rank 0 needs the frame, every peer has its own xGMI link to
// rank 0 (7 x ~153 GB/s into the root), and a ring all-gather would push 7/8 of the frame through EVERY link.
// A 4K fp32 frame is 99.5 MB, 12.4 MB per rank: ~0.1 ms per link when all transmit at once - against a render
// of 100+ ms per rank; there is nothing to overlap it with, so it is not chunked either.
//
// librccl.so (570 MB of code objects) is dlopen()ed when the first group is created, so that the single-device
// `rrt` never pays for it; inside a Python process that imported torch the already-mapped copy is reused, as
// for the HIP runtime itself (rrt_amd/_lib.py).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rrtx.h"
#include "rrtx_launch.h"

namespace rrtx {
int set_error(int code, const std::string &msg); // rrtx_api.cpp: the thread-local text behind rrtx_last_error()
}

namespace {

using namespace rrtx;

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

// (one attempt per process; the result - also a failure - is kept)
const Rccl *load_rccl(std::string &why)
{
    static std::mutex mu;
    static Rccl lib;
    static bool tried = false;
    static std::string error;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib.handle) break;
            error = dlerror();
        }
        if (lib.handle) {
            bool ok = true;
            auto sym = [&](const char *n) -> void * {
                void *p = dlsym(lib.handle, n);
                if (!p) ok = false, error = std::string("librccl: missing symbol ") + n;
                return p;
            };
            lib.CommInitAll = (decltype(lib.CommInitAll))sym("ncclCommInitAll");
            lib.CommDestroy = (decltype(lib.CommDestroy))sym("ncclCommDestroy");
            lib.GroupStart = (decltype(lib.GroupStart))sym("ncclGroupStart");
            lib.GroupEnd = (decltype(lib.GroupEnd))sym("ncclGroupEnd");
            lib.Send = (decltype(lib.Send))sym("ncclSend");
            lib.Recv = (decltype(lib.Recv))sym("ncclRecv");
            lib.GetErrorString = (decltype(lib.GetErrorString))sym("ncclGetErrorString");
            lib.GetVersion = (decltype(lib.GetVersion))sym("ncclGetVersion");
            if (!ok) {
                dlclose(lib.handle);
                lib.handle = nullptr;
            }
        }
    }
    if (!lib.handle) {
        why = "cannot load RCCL: " + error;
        return nullptr;
    }
    return &lib;
}

#define GRP_HIP(expr)                                                                                                    \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess) {                                                                                          \
            char buf_[512];                                                                                              \
            snprintf(buf_, sizeof buf_, "HIP error = %u (%s) at %s:%d '%s'", (unsigned)e_, hipGetErrorString(e_), __FILE__, \
                     __LINE__, #expr);                                                                                   \
            return set_error(RRTX_E_DEVICE, buf_);                                                                       \
        }                                                                                                                \
    } while (0)

#define GRP_NCCL(g, expr)                                                                                                \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess) {                                                                                         \
            char buf_[512];                                                                                              \
            snprintf(buf_, sizeof buf_, "RCCL error = %d (%s) at %s:%d '%s'", (int)r_, (g)->rccl->GetErrorString(r_), __FILE__, __LINE__, #expr); \
            return set_error(RRTX_E_DEVICE, buf_);                                                                       \
        }                                                                                                                \
    } while (0)

} // namespace

struct rrtx_group {
    rrtx_params p;
    int n = 0;
    bool rehearsal = false; // members share devices: no communicator, device-to-device copies instead
    std::vector<int> devs;
    std::vector<rrtx_ctx *> ctx;
    std::vector<void *> d_local;   // per member: its compact row block
    std::vector<uint32_t> rows;    // per member: rows in its shard
    std::vector<uint32_t> row_off; // rows before member r in the gathered buffer
    std::vector<hipEvent_t> ev_done;
    const Rccl *rccl = nullptr;
    int rccl_version = 0; // ncclGetVersion (0: rehearsal, no communicator)
    std::vector<ncclComm_t> comms;
    void *d_gathered = nullptr, *d_frame = nullptr; // on devs[0]
    hipEvent_t ev_begin = nullptr, ev_frame = nullptr;
    size_t fsize = 4;
    bool have_scene = false;
};

namespace {

int group_render_enqueue_and_wait(rrtx_group *g, rrtx_group_stats *stats, bool copy_back, void *fb)
{
    const auto t0 = std::chrono::steady_clock::now();
    const size_t row_bytes = (size_t)g->p.image_width * 3 * g->fsize;
    hipStream_t s0 = (hipStream_t)rrtx_stream(g->ctx[0]);
    GRP_HIP(hipSetDevice(g->devs[0]));
    GRP_HIP(hipEventRecord(g->ev_begin, s0));
    // ---- the N shard renders: enqueued, not waited for
    for (int r = 0; r < g->n; ++r) {
        if (g->rows[r] == 0) continue;
        int rc = rrtx_render_device(g->ctx[r], g->d_local[r], rrtx_stream(g->ctx[r]));
        if (rc) return rc;
    }
    // ---- ONE exchange: every rank's block to rank 0
    const ncclDataType_t dtype = g->p.fp64 ? ncclDouble : ncclFloat;
    if (!g->rehearsal) {
        GRP_NCCL(g, g->rccl->GroupStart());
        ncclResult_t bad = ncclSuccess; // (a group that was started is always ended, also when a call inside it fails)
        const char *where = "";
        for (int r = 0; r < g->n && bad == ncclSuccess; ++r) {
            if (g->rows[r] == 0) continue;
            const size_t count = (size_t)g->rows[r] * g->p.image_width * 3;
            // (rank 0 sends to itself like everyone else: one code path for every N, N = 1 included)
            bad = g->rccl->Send(g->d_local[r], count, dtype, 0, g->comms[r], (hipStream_t)rrtx_stream(g->ctx[r]));
            where = "ncclSend";
            if (bad != ncclSuccess) break;
            bad = g->rccl->Recv((unsigned char *)g->d_gathered + (size_t)g->row_off[r] * row_bytes, count, dtype, r, g->comms[0], s0);
            where = "ncclRecv";
        }
        const ncclResult_t ended = g->rccl->GroupEnd();
        if (bad == ncclSuccess && ended != ncclSuccess) bad = ended, where = "ncclGroupEnd";
        if (bad != ncclSuccess) return set_error(RRTX_E_DEVICE, std::string("RCCL error in ") + where + ": " + g->rccl->GetErrorString(bad));
    }
    else {
        for (int r = 0; r < g->n; ++r) {
            if (g->rows[r] == 0) continue;
            GRP_HIP(hipSetDevice(g->devs[r]));
            GRP_HIP(hipEventRecord(g->ev_done[r], (hipStream_t)rrtx_stream(g->ctx[r])));
            GRP_HIP(hipSetDevice(g->devs[0]));
            GRP_HIP(hipStreamWaitEvent(s0, g->ev_done[r], 0));
            GRP_HIP(hipMemcpyAsync((unsigned char *)g->d_gathered + (size_t)g->row_off[r] * row_bytes, g->d_local[r], (size_t)g->rows[r] * row_bytes, hipMemcpyDeviceToDevice, s0));
        }
    }
    // ---- rows to their places
    GRP_HIP(hipSetDevice(g->devs[0]));
    GatherShape S = {};
    S.row_values = (uint32_t)g->p.image_width * 3u, S.height = (uint32_t)g->p.image_height, S.tile_rows = (uint32_t)g->p.tile_rows, S.n_shards = (uint32_t)g->n;
    for (int r = 0; r < g->n; ++r) S.row_off[r] = g->row_off[r];
    if (g->p.fp64)
        GRP_HIP(launch_deinterleave<double>((const double *)g->d_gathered, (double *)g->d_frame, S, s0));
    else
        GRP_HIP(launch_deinterleave<float>((const float *)g->d_gathered, (float *)g->d_frame, S, s0));
    GRP_HIP(hipEventRecord(g->ev_frame, s0));
    if (copy_back) GRP_HIP(hipMemcpyAsync(fb, g->d_frame, row_bytes * (size_t)g->p.image_height, hipMemcpyDeviceToHost, s0));
    GRP_HIP(hipStreamSynchronize(s0));
    for (int r = 1; r < g->n; ++r) { // (their sends have completed - rank 0 received them - but their events are read below)
        GRP_HIP(hipSetDevice(g->devs[r]));
        GRP_HIP(hipStreamSynchronize((hipStream_t)rrtx_stream(g->ctx[r])));
    }
    const auto t1 = std::chrono::steady_clock::now();
    rrtx_group_stats st;
    memset(&st, 0, sizeof st);
    st.n_devices = g->n, st.rccl = g->rehearsal ? 0 : 1;
    st.accel_exact = 1;
    st.rccl_version = g->rccl_version, st.rccl_comms = (int32_t)g->comms.size();
    for (int r = 0; r < g->n && r < 16; ++r) st.devices[r] = g->devs[r];
    for (int r = 0; r < g->n; ++r) {
        rrtx_stats ms;
        int rc = rrtx_collect(g->ctx[r], &ms);
        if (rc) return rc;
        if (r < 16) st.kernel_ms[r] = ms.kernel_ms;
        if (ms.kernel_ms > st.render_ms) st.render_ms = ms.kernel_ms;
        st.samples += ms.samples, st.segments += ms.segments, st.prim_tests += ms.prim_tests, st.bytes_algorithmic += ms.bytes_algorithmic;
        st.gathered_bytes += (uint64_t)g->rows[r] * row_bytes;
        st.sample_chunk = ms.sample_chunk, st.accel_cells = ms.accel_cells;
        if (!ms.accel_exact) st.accel_exact = 0;
    }
    GRP_HIP(hipSetDevice(g->devs[0]));
    float ms = 0.f;
    GRP_HIP(hipEventElapsedTime(&ms, g->ev_begin, g->ev_frame));
    st.device_ms = ms;
    st.gather_ms = st.device_ms > st.render_ms ? st.device_ms - st.render_ms : 0.0;
    st.wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (stats) *stats = st;
    return RRTX_OK;
}

// A failure anywhere after the first shard render was enqueued must not leave the members with work in flight and
// undrained event slots (they would fail, or time garbage, on their next render): wait for every member's stream and
// collect its events, keeping the first failure's message.
int group_render_device(rrtx_group *g, rrtx_group_stats *stats, bool copy_back, void *fb)
{
    if (!g->have_scene) return set_error(RRTX_E_NO_SCENE, "rrtx_group_render: no scene set");
    const int rc = group_render_enqueue_and_wait(g, stats, copy_back, fb);
    if (rc != RRTX_OK) {
        const std::string keep = rrtx_last_error();
        for (int r = 0; r < g->n; ++r) {
            (void)hipSetDevice(g->devs[r]);
            (void)hipStreamSynchronize((hipStream_t)rrtx_stream(g->ctx[r]));
            (void)rrtx_collect(g->ctx[r], nullptr);
        }
        (void)hipSetDevice(g->devs[0]);
        set_error(rc, keep);
    }
    return rc;
}

} // namespace

extern "C" {

void rrtx_group_destroy(rrtx_group *g)
{
    if (!g) return;
    for (size_t r = 0; r < g->comms.size(); ++r)
        if (g->comms[r] && g->rccl) (void)g->rccl->CommDestroy(g->comms[r]);
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        if (r < (int)g->devs.size()) (void)hipSetDevice(g->devs[r]);
        if (r < (int)g->d_local.size() && g->d_local[r]) (void)hipFree(g->d_local[r]);
        if (r < (int)g->ev_done.size() && g->ev_done[r]) (void)hipEventDestroy(g->ev_done[r]);
        rrtx_destroy(g->ctx[r]);
    }
    if (!g->devs.empty()) (void)hipSetDevice(g->devs[0]);
    if (g->d_gathered) (void)hipFree(g->d_gathered);
    if (g->d_frame) (void)hipFree(g->d_frame);
    if (g->ev_begin) (void)hipEventDestroy(g->ev_begin);
    if (g->ev_frame) (void)hipEventDestroy(g->ev_frame);
    delete g;
}

int rrtx_group_create(const rrtx_params *params, int n_devices, const int32_t *devices, int flags, rrtx_group **out)
{
    if (!params || !out) return set_error(RRTX_E_INVALID, "rrtx_group_create: null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > kMaxGroup) return set_error(RRTX_E_INVALID, "rrtx_group_create: between 1 and 64 devices");
    int ndev = 0;
    GRP_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return set_error(RRTX_E_DEVICE, "rrtx_group_create: no HIP device present (this path has no CPU fallback)");
    rrtx_group *g = new rrtx_group();
    g->p = *params;
    g->n = n_devices;
    if (g->p.tile_rows < 1) g->p.tile_rows = 4;
    g->fsize = g->p.fp64 ? 8 : 4;
    bool dup = false;
    for (int r = 0; r < n_devices; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= ndev) {
            delete g;
            return set_error(RRTX_E_INVALID, "rrtx_group_create: device ordinal out of range (" + std::to_string(d) + " of " + std::to_string(ndev) + ")");
        }
        for (int q : g->devs) dup = dup || q == d;
        g->devs.push_back(d);
    }
    if (dup && !(flags & RRTX_GROUP_REHEARSAL)) {
        delete g;
        return set_error(RRTX_E_INVALID, "rrtx_group_create: a device is listed twice (only a rehearsal may share devices: RRTX_GROUP_REHEARSAL)");
    }
    g->rehearsal = dup;
    auto bail = [&](int rc) {
        rrtx_group_destroy(g);
        return rc;
    };
    uint32_t off = 0;
    for (int r = 0; r < n_devices; ++r) {
        rrtx_params pr = g->p;
        pr.device = g->devs[r], pr.shard_rank = r, pr.shard_count = n_devices;
        rrtx_ctx *c = nullptr;
        int rc = rrtx_create(&pr, &c);
        if (rc) return bail(rc);
        g->ctx.push_back(c);
        const int rows = rrtx_shard_rows(c, nullptr, 0);
        g->rows.push_back((uint32_t)rows), g->row_off.push_back(off);
        off += (uint32_t)rows;
        void *d = nullptr;
        hipEvent_t ev = nullptr;
        hipError_t e = hipSetDevice(g->devs[r]);
        if (e == hipSuccess) e = hipMalloc(&d, (size_t)rows * g->p.image_width * 3 * g->fsize + 64);
        g->d_local.push_back(d);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        g->ev_done.push_back(ev);
        if (e != hipSuccess) return bail(set_error(RRTX_E_DEVICE, std::string("rrtx_group_create: ") + hipGetErrorString(e)));
    }
    {
        const size_t frame_bytes = (size_t)g->p.image_height * g->p.image_width * 3 * g->fsize;
        hipError_t e = hipSetDevice(g->devs[0]);
        if (e == hipSuccess) e = hipMalloc(&g->d_gathered, frame_bytes + 64);
        if (e == hipSuccess) e = hipMalloc(&g->d_frame, frame_bytes + 64);
        if (e == hipSuccess) e = hipEventCreate(&g->ev_begin);
        if (e == hipSuccess) e = hipEventCreate(&g->ev_frame);
        if (e != hipSuccess) return bail(set_error(RRTX_E_DEVICE, std::string("rrtx_group_create: ") + hipGetErrorString(e)));
    }
    if (!g->rehearsal) {
        std::string why;
        g->rccl = load_rccl(why);
        if (!g->rccl) return bail(set_error(RRTX_E_DEVICE, "rrtx_group_create: " + why));
        g->comms.assign(n_devices, nullptr);
        // (RCCL greets on STDOUT, "RCCL version : ...".  Round 3 pointed file descriptor 1 at stderr here while the communicators were
        // built - a process-wide side effect a library has no business with; since ABI 4 the hosts that own stdout do it themselves:
        // rrt_main.cpp around its rrtx_group_create, rrt_amd/render.py's RrtGroup.  RRTX_GROUP_KEEP_STDOUT is accepted and ignored.)
        ncclResult_t r = g->rccl->CommInitAll(g->comms.data(), n_devices, g->devs.data());
        if (r != ncclSuccess) return bail(set_error(RRTX_E_DEVICE, std::string("rrtx_group_create: ncclCommInitAll: ") + g->rccl->GetErrorString(r)));
        int v = 0;
        if (g->rccl->GetVersion(&v) == ncclSuccess) g->rccl_version = v;
    }
    *out = g;
    return RRTX_OK;
}

int rrtx_group_size(const rrtx_group *g) { return g ? g->n : 0; }

rrtx_ctx *rrtx_group_member(rrtx_group *g, int i) { return g && i >= 0 && i < g->n ? g->ctx[i] : nullptr; }

int rrtx_group_set_scene(rrtx_group *g, const rrtx_scene_desc *scene)
{
    if (!g || !scene) return set_error(RRTX_E_INVALID, "rrtx_group_set_scene: null argument");
    g->have_scene = false;
    for (int r = 0; r < g->n; ++r) {
        int rc = rrtx_set_scene(g->ctx[r], scene);
        if (rc) return rc;
    }
    g->have_scene = true;
    return RRTX_OK;
}

int rrtx_group_render(rrtx_group *g, void *fb, rrtx_group_stats *stats)
{
    if (!g || !fb) return set_error(RRTX_E_INVALID, "rrtx_group_render: null argument");
    return group_render_device(g, stats, true, fb);
}

int rrtx_group_render_device(rrtx_group *g, void **d_frame, rrtx_group_stats *stats)
{
    if (!g || !d_frame) return set_error(RRTX_E_INVALID, "rrtx_group_render_device: null argument");
    int rc = group_render_device(g, stats, false, nullptr);
    *d_frame = rc ? nullptr : g->d_frame;
    return rc;
}

} // extern "C"
