// C ABI of the render path (include/rrtx.h): context, scene packing/upload, launches, timing.
// The host half of Rrt::render (rrt.cu:186-334), without the device heap, the 1-thread
// create_world kernel or the per-pixel curand state.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/rrtx.h"
#include "rrtx_grid.h"
#include "rrtx_pack.h"
#include "rrtx_launch.h"

namespace {

using namespace rrtx;

thread_local std::string g_error;

int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}

#define RRTX_HIP(expr)                                                                                                   \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess) {                                                                                          \
            char buf_[512];                                                                                              \
            snprintf(buf_, sizeof buf_, "HIP error = %u (%s) at %s:%d '%s'", (unsigned)e_, hipGetErrorString(e_), __FILE__, \
                     __LINE__, #expr);                                                                                   \
            return fail(RRTX_E_DEVICE, buf_);                                                                            \
        }                                                                                                                \
    } while (0)

constexpr int kEventRing = 64;
constexpr size_t kFirstBounceMinSamples = (size_t)16 << 20; // see set_scene

} // namespace

namespace rrtx {
int set_error(int code, const std::string &msg) { return fail(code, msg); } // for the other translation units of the library (rrtx_group.cpp)
}

struct rrtx_ctx {
    rrtx_params p;
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int blocks_per_cu = 0;
    // derived geometry of this shard
    int local_rows = 0;
    int chunk = 0, chunks_per_pixel = 0;
    int sum_chunk = 0;        // the summation shape the image has (rrtx_stats.sample_chunk): == chunk, except in per-sample launches, where the tasks (chunk) are small
    bool per_sample = false;  // every sample's radiance stored by itself, summed by finalize_kernel in chunks of sum_chunk (sample_chunk = -1 at spp > 16: the reference's order)
    uint32_t total_tasks = 0;
    uint32_t taper_pixel = 0; // first local pixel cut into single-sample tasks
    bool use_partial = false; // tasks write partial sums, finalize_kernel forms the frame
    size_t fsize = 4;
    // scene
    bool have_scene = false;
    int n_sph = 0, n_sph_padded = 0, n_msph = 0, n_tri = 0, n_mat = 0;
    void *d_hot = nullptr, *d_filter = nullptr, *d_cold = nullptr, *d_msph = nullptr, *d_tri = nullptr, *d_tri_scan = nullptr, *d_mat = nullptr;
    bool tail_ok = false;    // scene magnitudes allow the tail kernel's split scan (no NaN roots possible)
    bool use_filter = false; // conservative scan filter valid for the current scene and not disabled
    int lds_mode = 0;        // 0 scalar loads, 1 alternate scalar / LDS, 2 LDS only, 3 the filter on the matrix cores (f16 operands in LDS)
    uint32_t *d_mf_table = nullptr, *d_mf_big = nullptr; // lds_mode 3: rrtx_pack.h, pack_mf_table
    int n_mf_big = 0;
    unsigned char cam_bytes[sizeof(CameraRec<double>)];
    // work buffers
    uint32_t *d_queue = nullptr;
    unsigned long long *d_counters = nullptr;
    void *d_partial = nullptr; // [total_tasks][3] unless every pixel is a single task
    uint16_t *d_plist = nullptr;  // camera-ray candidate lists [local pixel][kPlistStride]
    bool have_plist = false;      // ... valid for the current scene (the buffer itself is kept from scene to scene)
    uint32_t *d_order = nullptr, *d_order_scratch = nullptr; // the sky split (KernelParams::pixel_order): the pixels with a non-empty list first ...
    uint32_t n_queue_pixels = 0;  // ... this many of them
    bool have_split = false;
    void *d_first = nullptr;      // the first-bounce records of a launch (KernelParams::first)
    bool have_first = false;
    // accelerated closest hit (use_bvh): uniform grid + always-list, see build_grid()
    bool accel = false;
    bool tail_grid = false;   // list scan (-b), but what a launch parks at its end is finished through the grid (a RESUME pass: same bits)
    uint32_t *d_grid_cell_start = nullptr, *d_grid_always = nullptr;
    GridPrim *d_grid_cell_prims = nullptr;
    bool accel_exact = true; // the grid is proven to reproduce the list scan bit for bit (false: fp32 triangles gridded under the approximate rule)
    int n_grid_cells = 0, n_grid_prims = 0, n_always = 0;
    unsigned char grid_bytes[sizeof(GridRec<double>)];
    void *d_tail_items = nullptr; // parked work items (render kernel -> tail kernel)
    void *d_tail_rad = nullptr;   // the results of their work units
    uint32_t *d_tail_units = nullptr;
#ifdef RRTX_DIAG
    unsigned long long *d_diag = nullptr;
#endif
    size_t tail_capacity = 0;
    int handoff_lanes = kHandoffLanes;
    int tail_blocks = 0;
    int resume_blocks = 0;    // grid of the resume pass (= the render grid in the accelerated mode; its own occupancy after a list-scan render)
    void *d_rows = nullptr;    // own output buffer for the host-pointer API
    std::vector<std::pair<void **, size_t>> caps; // capacities of the per-scene device buffers (see ensure_buffer)
    // timing
    hipEvent_t ev_start[kEventRing], ev_stop[kEventRing];
    // the sky split's kernel runs BESIDE the render kernel, on a stream of its own of the lowest priority: the persistent render kernel fills the device, the
    // sky kernel's blocks move in as its waves retire - into the end of the launch, where a handful of 50-bounce paths are all that is left to do
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int ev_pending = 0;
    double kernel_ms_total = 0.0;
    long renders_total = 0;
    int grid_blocks = 0;
    double last_wall_ms = 0.0;
};

namespace {

int rows_of_shard(const rrtx_params &p, std::vector<int32_t> *out)
{
    // (the plan's one statement: rrtx_device.h, held against its definition - row j belongs to shard (j / T) mod N - by tests/gather_plan_check.cpp)
    const uint32_t n = p.shard_count < 1 ? 1u : (uint32_t)p.shard_count, T = p.tile_rows < 1 ? 1u : (uint32_t)p.tile_rows;
    const uint32_t count = shard_row_count((uint32_t)p.image_height, T, n, (uint32_t)p.shard_rank);
    if (out)
        for (uint32_t lr = 0; lr < count; ++lr) out->push_back((int32_t)shard_local_to_frame_row(lr, T, n, (uint32_t)p.shard_rank));
    return (int)count;
}

// The per-scene device buffers (tables, grid, camera-ray lists, parked-item buffers: 200 MB at 1280x720) survive from scene
// to scene and are reallocated only when one has to grow: a batch of frames used to pay ~15 hipMalloc / hipFree pairs per
// scene (0.97 ms per rrtx_set_scene, of a 5.2 ms render at the reference's animation settings).
int ensure_buffer(rrtx_ctx *c, void **slot, size_t bytes)
{
    size_t *cap = nullptr;
    for (auto &e : c->caps)
        if (e.first == slot) cap = &e.second;
    if (!cap) {
        c->caps.push_back({slot, 0});
        cap = &c->caps.back().second;
    }
    if (*slot && *cap >= bytes) return RRTX_OK;
    if (*slot) (void)hipFree(*slot);
    *slot = nullptr, *cap = 0;
    const size_t want = bytes + bytes / 4 + 256; // (room to grow: the next frame of an animation has a few primitives more or less)
    RRTX_HIP(hipMalloc(slot, want));
    *cap = want;
    return RRTX_OK;
}

template <typename F> int upload_scene(rrtx_ctx *c, const rrtx_scene_desc *s)
{
    PackedScene<F> packed;
    if (const char *what = pack_scene<F>(s, packed)) return fail(RRTX_E_INVALID, what);
    memcpy(c->cam_bytes, &packed.cam, sizeof(CameraRec<F>));
    std::vector<MaterialRec<F>> &hmat = packed.mat;
    std::vector<SphereHot<F>> &hhot = packed.hot;
    std::vector<SphereHot<float>> &hfil = packed.filter;
    std::vector<SphereCold<F>> &hcold = packed.cold;
    std::vector<MovingSphereRec<F>> &hms = packed.ms;
    std::vector<TriangleRec<F>> &htri = packed.tri;
    const int n_pad = packed.n_pad;
    const bool filter_ok = packed.filter_ok;

    RRTX_HIP(hipSetDevice(c->device));
    c->have_scene = false;

    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        if (int e = ensure_buffer(c, dst, bytes)) return e;
        if (bytes) RRTX_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up(&c->d_hot, hhot.data(), hhot.size() * sizeof(SphereHot<F>)))) return rc;
    if ((rc = up(&c->d_filter, hfil.data(), hfil.size() * sizeof(SphereHot<float>)))) return rc;
    c->use_filter = filter_ok && !(c->p.flags & RRTX_FLAG_EXACT_SCAN);
    c->lds_mode = 0;
    if (c->use_filter && hfil.size() * sizeof(SphereHot<float>) <= (size_t)kLdsSceneBytes && !(c->p.flags & RRTX_FLAG_SCAN_SCALAR_ONLY))
        c->lds_mode = (c->p.flags & RRTX_FLAG_SCAN_LDS_ONLY) ? 2 : 1; // (fp64 too: its filter is the fp32 one)
    // scenes of spheres alone whose f16 operands fit LDS (one copy per block): the filter as two chained v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays
    if (c->lds_mode == 1 && packed.tail_ok && s->num_spheres > 0 && s->num_moving_spheres == 0 && s->num_triangles == 0 && (size_t)mf_padded(n_pad) * 64 <= (size_t)kLdsMfBytes && !(c->p.flags & (RRTX_FLAG_SCAN_NO_MFMA | RRTX_FLAG_VERIFY_LISTS))) {
        MfTable mf;
        pack_mf_table<F>(hhot, s->num_spheres, n_pad, mf);
        if (mf.ok) {
            c->n_mf_big = (int)mf.big.size();
            if (mf.big.empty()) mf.big.push_back(0); // (no empty uploads)
            if ((rc = up((void **)&c->d_mf_table, mf.halves.data(), mf.halves.size() * 2))) return rc;
            if ((rc = up((void **)&c->d_mf_big, mf.big.data(), mf.big.size() * 4))) return rc;
            c->lds_mode = 3;
        }
    }
    if ((rc = up(&c->d_cold, hcold.data(), hcold.size() * sizeof(SphereCold<F>)))) return rc;
    if ((rc = up(&c->d_msph, hms.data(), hms.size() * sizeof(MovingSphereRec<F>)))) return rc;
    if ((rc = up(&c->d_tri, htri.data(), htri.size() * sizeof(TriangleRec<F>)))) return rc;
    if ((rc = up(&c->d_tri_scan, packed.tri_scan.data(), packed.tri_scan.size() * sizeof(TriScanRec<F>)))) return rc;
    if ((rc = up(&c->d_mat, hmat.data(), hmat.size() * sizeof(MaterialRec<F>)))) return rc;
    c->tail_ok = packed.tail_ok;
    // accelerated closest hit (the reference's -b switches its BVH off, main.cpp:67,90)
    c->accel = false, c->tail_grid = false;
    // (also without use_bvh: the paths a list-scan launch parks at its end - half a percent of its segments - are then
    // finished lane per ray through the grid instead of 8 lanes per ray on the list: 0.75 against 1.25 ms for an eighth of the
    // spp 500 frame, and the same bits - only where the grid is PROVEN to give them; RRTX_FLAG_NO_TAIL_GRID: A/B switch)
    if (c->tail_ok && !(c->p.flags & RRTX_FLAG_EXACT_SCAN) && (c->p.use_bvh || !(c->p.flags & (RRTX_FLAG_NO_TAIL_GRID | RRTX_FLAG_NO_TAIL_KERNEL | RRTX_FLAG_VERIFY_LISTS)))) {
        std::vector<uint32_t> cell_start, always;
        std::vector<GridPrim> cell_prims;
        GridRec<F> G = {};
        CameraRec<F> camrec;
        memcpy(&camrec, c->cam_bytes, sizeof camrec);
        // Triangles whose test the grid is not PROVEN to cover (fp32: practically all, rrtx_grid.h) are gridded all the same,
        // under an empirical inflation, unless the caller insists on the list scan's bits (RRTX_FLAG_EXACT_ACCEL): the
        // reference's own default, its BVH, has the same hazard band against its own list scan (bvh.h:167-175)
        bool approximate = false;
        bool built = build_grid<F>(hhot, hcold, s->num_spheres, n_pad, hms, s->num_moving_spheres, htri, s->num_triangles, camrec, cell_start, cell_prims, always, G,
                                   c->p.use_bvh && !(c->p.flags & RRTX_FLAG_EXACT_ACCEL), &approximate);
        // (a handful of primitives: no grid, the list is scanned also under use_bvh - but the END of the launch still goes
        // through a grid of nothing but an always-list: build_degenerate_grid)
        bool degenerate = false;
        if (!built && !(c->p.flags & (RRTX_FLAG_NO_TAIL_GRID | RRTX_FLAG_NO_TAIL_KERNEL | RRTX_FLAG_VERIFY_LISTS)))
            built = degenerate = build_degenerate_grid<F>(s->num_spheres, n_pad, s->num_moving_spheres, s->num_triangles, cell_start, cell_prims, always, G);
        if (built) {
            c->accel_exact = !approximate;
            if (cell_prims.empty()) cell_prims.push_back(0);
            if (always.empty()) always.push_back(0), c->n_always = 0;
            else c->n_always = (int)always.size();
            if ((rc = up((void **)&c->d_grid_cell_start, cell_start.data(), cell_start.size() * 4))) return rc;
            if ((rc = up((void **)&c->d_grid_cell_prims, cell_prims.data(), cell_prims.size() * sizeof(GridPrim)))) return rc;
            if ((rc = up((void **)&c->d_grid_always, always.data(), always.size() * 4))) return rc;
            c->n_grid_cells = (int)G.dims[0] * (int)G.dims[1] * (int)G.dims[2]; // (cell_start may carry the coarse occupancy bytes behind its cells + 1 words)
            c->n_grid_prims = (int)cell_start[(size_t)c->n_grid_cells];
            memcpy(c->grid_bytes, &G, sizeof G);
            c->accel = c->p.use_bvh != 0 && !degenerate;
            c->tail_grid = !c->accel;
        }
    }
    c->n_sph = s->num_spheres;
    c->n_sph_padded = n_pad;
    c->n_msph = s->num_moving_spheres;
    c->n_tri = s->num_triangles;
    c->n_mat = s->num_materials;
    c->have_scene = true;
    return RRTX_OK;
}

// with_grid: the acceleration grid's tables go along (accelerated render and resume passes; a list-scan render takes none)
template <typename F> KernelParams<F> make_params(const rrtx_ctx *c, void *out, bool with_grid)
{
    KernelParams<F> P = {};
    P.sph_hot = (const SphereHot<F> *)c->d_hot;
    P.sph_filter = (const SphereHot<float> *)c->d_filter;
    P.mf_table = c->d_mf_table, P.mf_big = c->d_mf_big, P.n_mf_big = c->lds_mode == 3 ? c->n_mf_big : 0;
    P.sph_cold = (const SphereCold<F> *)c->d_cold;
    P.msph = (const MovingSphereRec<F> *)c->d_msph;
    P.tri = (const TriangleRec<F> *)c->d_tri;
    P.tri_scan = (const TriScanRec<F> *)c->d_tri_scan;
    P.mat = (const MaterialRec<F> *)c->d_mat;
    P.n_sph = c->n_sph, P.n_sph_padded = c->n_sph_padded, P.n_msph = c->n_msph, P.n_tri = c->n_tri;
    memcpy(&P.cam, c->cam_bytes, sizeof(CameraRec<F>));
    P.W = c->p.image_width, P.H = c->p.image_height, P.spp = c->p.samples_per_pixel, P.max_depth = c->p.max_depth;
    P.seed = c->p.seed;
    P.chunk = c->chunk, P.chunks_per_pixel = c->chunks_per_pixel, P.per_sample = c->per_sample ? 1 : 0;
    P.local_rows = c->local_rows;
    P.tile_rows = c->p.tile_rows, P.shard_rank = c->p.shard_rank, P.shard_count = c->p.shard_count;
    P.taper_pixel = c->taper_pixel, P.taper_task_base = c->taper_pixel * (uint32_t)c->chunks_per_pixel;
    P.total_tasks = c->total_tasks;
    P.div_cpp = make_fastdiv((uint32_t)c->chunks_per_pixel), P.div_spp = make_fastdiv((uint32_t)c->p.samples_per_pixel);
    P.div_w = make_fastdiv((uint32_t)c->p.image_width), P.div_tile = make_fastdiv((uint32_t)c->p.tile_rows);
    P.queue = c->d_queue;
    P.out = (F *)out;
    P.counters = c->d_counters;
    P.collect_stats = c->p.collect_stats;
    P.handoff_lanes = c->tail_capacity ? c->handoff_lanes : 0;
    P.handoff_iters = c->p.handoff_iters > 0 ? c->p.handoff_iters : ((c->accel && (c->n_msph > 0 || c->n_tri > 0)) ? kHandoffItersDense : kHandoffIters); // (dense: the variants that pair (ray, entry) across the wave)
    P.tail_count = c->d_queue + 1;
    P.tail_items = (TailItem<F> *)c->d_tail_items;
    P.tail_rad = (F *)c->d_tail_rad;
    P.tail_units = c->d_tail_units;
    if (with_grid && (c->accel || c->tail_grid)) {
        P.grid_cell_start = c->d_grid_cell_start, P.grid_cell_prims = c->d_grid_cell_prims, P.grid_always = c->d_grid_always;
        P.n_always = c->n_always, P.n_grid_cells = c->n_grid_cells, P.n_grid_prims = c->n_grid_prims;
        memcpy(&P.grid, c->grid_bytes, sizeof(GridRec<F>));
    }
    P.plist = c->have_plist ? c->d_plist : nullptr;
    P.first = c->have_first ? c->d_first : nullptr;
    if (c->have_split) {
        P.pixel_order = c->d_order, P.n_queue_pixels = c->n_queue_pixels;
        P.total_tasks = c->n_queue_pixels * (uint32_t)c->chunks_per_pixel; // the render kernel's queue; sky_tasks_kernel takes the positions behind it
    }
    P.list_passes = c->have_plist ? (c->p.list_passes > 0 ? c->p.list_passes : (c->p.list_passes < 0 ? 0 : kListPasses)) : 0;
    P.verify_lists = (c->p.flags & RRTX_FLAG_VERIFY_LISTS) ? 1 : 0;
    P.diag = nullptr;
#ifdef RRTX_DIAG
    P.diag = c->d_diag;
#endif
    return P;
}

// Everything rrtx_set_scene allocates (tables, grid, camera-ray lists, parked-item buffers).
void free_scene_buffers(rrtx_ctx *c)
{
    void **bufs[] = {&c->d_hot, &c->d_filter, &c->d_cold, &c->d_msph, &c->d_tri, &c->d_tri_scan, &c->d_mat, (void **)&c->d_grid_cell_start, (void **)&c->d_grid_cell_prims,
                     (void **)&c->d_grid_always, (void **)&c->d_mf_table, (void **)&c->d_mf_big, (void **)&c->d_plist, (void **)&c->d_order, (void **)&c->d_order_scratch, &c->d_first, &c->d_tail_items, &c->d_tail_rad, (void **)&c->d_tail_units};
    for (void **b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    c->caps.clear();
    c->have_scene = c->have_plist = c->have_split = c->have_first = false;
    c->accel = c->tail_grid = false;
    c->tail_capacity = 0;
}

int drain_events(rrtx_ctx *c, double *last_ms)
{
    for (int i = 0; i < c->ev_pending; ++i) {
        RRTX_HIP(hipEventSynchronize(c->ev_stop[i]));
        float ms = 0.f;
        RRTX_HIP(hipEventElapsedTime(&ms, c->ev_start[i], c->ev_stop[i]));
        c->kernel_ms_total += ms;
        c->renders_total += 1;
        if (last_ms) *last_ms = ms;
    }
    c->ev_pending = 0;
    return RRTX_OK;
}

} // namespace

extern "C" {

const char *rrtx_version(void) { return RRTX_VERSION_STRING; }
int rrtx_abi_version(void) { return RRTX_ABI_VERSION; }
const char *rrtx_last_error(void) { return g_error.c_str(); }

int rrtx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rrtx_runtime_version(void)
{
    int v = -1;
    if (hipRuntimeGetVersion(&v) != hipSuccess) return -1;
    return v;
}

int rrtx_pin_host(void *ptr, size_t bytes)
{
    if (!ptr || bytes == 0) return RRTX_OK;
    RRTX_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return RRTX_OK;
}

int rrtx_unpin_host(void *ptr)
{
    if (!ptr) return RRTX_OK;
    RRTX_HIP(hipHostUnregister(ptr));
    return RRTX_OK;
}

int rrtx_query(int device, rrtx_devinfo *out)
{
    if (!out) return fail(RRTX_E_INVALID, "rrtx_query: null output");
    hipDeviceProp_t prop;
    RRTX_HIP(hipGetDeviceProperties(&prop, device));
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", prop.name);
    out->major = prop.major, out->minor = prop.minor;
    out->multi_processor_count = prop.multiProcessorCount;
    out->shared_mem_per_block = (int64_t)prop.sharedMemPerBlock;
    out->max_threads_per_block = prop.maxThreadsPerBlock;
    out->max_threads_per_multiprocessor = prop.maxThreadsPerMultiProcessor;
    out->unified_addressing = 1;
    out->l2_cache_size = prop.l2CacheSize;
    out->total_global_mem = (int64_t)prop.totalGlobalMem;
    out->clock_khz = prop.clockRate;
    return RRTX_OK;
}

int rrtx_create(const rrtx_params *params, rrtx_ctx **out)
{
    if (!params || !out) return fail(RRTX_E_INVALID, "rrtx_create: null argument");
    *out = nullptr;
    rrtx_params p = *params;
    if (p.image_width < 2 || p.image_height < 2) return fail(RRTX_E_INVALID, "rrtx_create: image must be at least 2x2 (u = (i+xi)/(w-1), rrt.cu:112)");
    if (p.samples_per_pixel < 1) return fail(RRTX_E_INVALID, "rrtx_create: samples_per_pixel must be >= 1");
    if ((int64_t)p.image_width * p.image_height > (int64_t)1 << 30) return fail(RRTX_E_INVALID, "rrtx_create: image too large");
    if (p.shard_count < 1) p.shard_count = 1;
    if (p.tile_rows < 1) p.tile_rows = 1;
    if (p.shard_rank < 0 || p.shard_rank >= p.shard_count) return fail(RRTX_E_INVALID, "rrtx_create: shard_rank out of range");
    if (p.seed == 0) p.seed = 1984u;

    int ndev = 0;
    RRTX_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(RRTX_E_DEVICE, "rrtx_create: no HIP device present (this path has no CPU fallback)");
    if (p.device < 0 || p.device >= ndev) return fail(RRTX_E_INVALID, "rrtx_create: device ordinal out of range");
    RRTX_HIP(hipSetDevice(p.device));

    rrtx_ctx *c = new rrtx_ctx();
    c->p = p;
    c->device = p.device;
    c->fsize = p.fp64 ? 8 : 4;
    c->local_rows = rows_of_shard(p, nullptr);

    // samples per work item: a function of (w, h, spp) only, so that the summation shape — and
    // with it the image — does not depend on how many devices share the frame.
    const int64_t pixels_full = (int64_t)p.image_width * p.image_height;
    int chunk = p.sample_chunk;
    if (chunk < 0 || chunk >= p.samples_per_pixel)
        chunk = p.samples_per_pixel;
    else if (chunk == 0) {
        chunk = p.samples_per_pixel > 8 ? 8 : p.samples_per_pixel;
        // 8 samples per work item; 16 once the FULL frame's partial sums (3 values per item) pass 2 GiB - per-item costs
        // (hand-out, decoding, the 12-byte store) then outweigh what shorter items save; more only where the items could
        // not be counted in 31 bits or their sums would pass 48 GiB, a sixth of one MI355X's HBM.  Long items cost at the
        // END of a launch, where every lane holds half an item when the queue runs dry: one of the 8 shards of the
        // 3840x2160 spp 1000 frame takes 207 ms with 128 samples per item (what the old rule - 2 GiB at most - chose),
        // 165 with 8 or 16, against 160 for an eighth of the frame; the full frame on one GPU: 1299 / 1292 / 1281 ms with
        // 8 / 16 / 128 (use_bvh: 684 / 657 / 641).  A function of (w, h, spp) only: the same bits on any number of GPUs.
        for (;;) {
            const int64_t cpp = (p.samples_per_pixel + chunk - 1) / chunk;
            const int64_t bytes = pixels_full * cpp * 3 * 8;
            const bool fits = pixels_full * cpp < ((int64_t)1 << 31) - (1 << 20) && bytes <= ((int64_t)48 << 30);
            if ((fits && (bytes <= ((int64_t)2 << 30) || chunk >= 16)) || chunk >= p.samples_per_pixel) break;
            chunk *= 2;
        }
        if (chunk > p.samples_per_pixel) chunk = p.samples_per_pixel;
    }
    // One work item per pixel (the reference's own order of summation: one running sum over all samples, rrt.cu:115) is a schedule of 500-sample items: twice
    // the time of 8-sample ones (98.5 against 48.9 ms on configuration 3).  The order does not need the schedule: the tasks stay small and write every sample's
    // radiance to its own slot ([pixel][sample][3]: 12 bytes a sample, 5.8 GB for 1200x800 spp 500), finalize_kernel adds a pixel's samples up in one running sum.
    c->sum_chunk = chunk;
    c->per_sample = chunk == p.samples_per_pixel && p.samples_per_pixel > 16 && !(p.flags & RRTX_FLAG_ONE_ITEM_PER_PIXEL) &&
                    (int64_t)c->local_rows * p.image_width * p.samples_per_pixel * 3 * (int64_t)c->fsize <= ((int64_t)96 << 30);
    if (c->per_sample) chunk = 8;
    c->chunk = chunk;
    c->chunks_per_pixel = (p.samples_per_pixel + chunk - 1) / chunk;
    // the last `taper` samples of the queue go out as single-sample tasks (task_decode in rrtx_kernels.hip)
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, p.device);
    const int64_t local_pixels = (int64_t)c->local_rows * p.image_width;
    int64_t taper = p.taper_samples > 0 ? p.taper_samples : (p.taper_samples < 0 || e != hipSuccess ? 0 : kTaperSamplesPerCu * prop.multiProcessorCount);
    if (chunk <= 1 || c->per_sample) taper = 0;
    int64_t tapered_pixels = (taper + p.samples_per_pixel - 1) / p.samples_per_pixel;
    if (tapered_pixels > local_pixels) tapered_pixels = local_pixels;
    const int64_t tasks = (local_pixels - tapered_pixels) * c->chunks_per_pixel + tapered_pixels * p.samples_per_pixel;
    if (tasks >= ((int64_t)1 << 31)) {
        delete c;
        return fail(RRTX_E_INVALID, "rrtx_create: too many work items; raise sample_chunk");
    }
    c->total_tasks = (uint32_t)tasks;
    c->taper_pixel = (uint32_t)(local_pixels - tapered_pixels);
    c->use_partial = c->chunks_per_pixel > 1 || tapered_pixels > 0;

    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    // (one allocation: the launch's control words - 256 bytes - and the statistics counters behind them are cleared by ONE fill per launch)
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_queue, 512);
    if (e == hipSuccess) c->d_counters = (unsigned long long *)(c->d_queue + 64);
    if (e == hipSuccess) e = hipMemset(c->d_queue, 0, 512);
    if (e == hipSuccess && c->use_partial) {
        // (footprint: one vec3 per work item - 0.73 GB for 1200x800 spp 500, 6.3 GB fp32 / 12.5 GB fp64 for 3840x2160 spp 1000 at
        // 16 samples per item; several contexts on one device each hold their own)
        const size_t bytes = (c->per_sample ? (size_t)local_pixels * (size_t)p.samples_per_pixel : (size_t)c->total_tasks) * 3 * c->fsize + 64;
        e = hipMalloc(&c->d_partial, bytes);
        if (e != hipSuccess) {
            char buf[384];
            snprintf(buf, sizeof buf, "rrtx_create: cannot allocate %.2f GB for the per-work-item sums (%u work items of %d samples): HIP error = %u (%s); raise sample_chunk (`rrt -C`) to shrink it",
                     (double)bytes / 1e9, (unsigned)c->total_tasks, c->chunk, (unsigned)e, hipGetErrorString(e));
            rrtx_destroy(c);
            return fail(RRTX_E_DEVICE, buf);
        }
    }
#ifdef RRTX_DIAG
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_diag, (size_t)131072 * 64);
    if (e == hipSuccess) e = hipMemset(c->d_diag, 0, (size_t)131072 * 64);
#endif
    for (int i = 0; i < kEventRing && e == hipSuccess; ++i) {
        e = hipEventCreate(&c->ev_start[i]);
        if (e == hipSuccess) e = hipEventCreate(&c->ev_stop[i]);
    }
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof buf, "rrtx_create: HIP error = %u (%s)", (unsigned)e, hipGetErrorString(e));
        rrtx_destroy(c);
        return fail(RRTX_E_DEVICE, buf);
    }
    c->num_cus = prop.multiProcessorCount;
    c->grid_blocks = 1;
    *out = c;
    return RRTX_OK;
}

void rrtx_destroy(rrtx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->ev_pending; ++i) (void)hipEventSynchronize(c->ev_stop[i]); // (renders enqueued on a caller's stream)
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    free_scene_buffers(c); // everything rrtx_set_scene allocated: tables (the matrix form's operands too), grid, lists, parked-item buffers
    void *bufs[] = {c->d_queue, c->d_partial, c->d_rows};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
#ifdef RRTX_DIAG
    if (c->d_diag) (void)hipFree(c->d_diag);
#endif
    for (int i = 0; i < kEventRing; ++i) {
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int set_scene_impl(rrtx_ctx *c, const rrtx_scene_desc *s);

int rrtx_set_scene(rrtx_ctx *c, const rrtx_scene_desc *s)
{
    if (!c || !s) return fail(RRTX_E_INVALID, "rrtx_set_scene: null argument");
    const int rc = set_scene_impl(c, s);
    if (rc != RRTX_OK) {
        // a scene that did not make it leaves the context without one - and without its buffers: every
        // device allocation of the attempt is owned by the context as soon as it exists, so this frees
        // them all (g_error keeps the first failure's message)
        const std::string keep = g_error;
        (void)hipSetDevice(c->device);
        free_scene_buffers(c);
        g_error = keep;
    }
    return rc;
}

static int set_scene_impl(rrtx_ctx *c, const rrtx_scene_desc *s)
{
    if ((s->fp64 != 0) != (c->p.fp64 != 0)) return fail(RRTX_E_INVALID, "rrtx_set_scene: scene precision differs from the context's");
    if (!s->camera) return fail(RRTX_E_INVALID, "rrtx_set_scene: no camera");
    if (s->num_materials < 1 || !s->materials) return fail(RRTX_E_INVALID, "rrtx_set_scene: no materials");
    if (s->num_spheres < 0 || s->num_moving_spheres < 0 || s->num_triangles < 0) return fail(RRTX_E_INVALID, "rrtx_set_scene: negative count");
    if ((s->num_spheres && !s->spheres) || (s->num_moving_spheres && !s->moving_spheres) || (s->num_triangles && !s->triangles))
        return fail(RRTX_E_INVALID, "rrtx_set_scene: count without table");
    if ((int64_t)s->num_spheres + s->num_moving_spheres + s->num_triangles > (1 << 28)) return fail(RRTX_E_INVALID, "rrtx_set_scene: too many primitives");
    // (the tables of the previous scene are overwritten in place: nothing may still be rendering from them, on whatever stream
    // the caller launched - rrtx_render_device takes any and records a stop event on it; waiting for those events and for the
    // context's own stream instead of the whole device leaves other contexts and a host application's streams alone)
    RRTX_HIP(hipSetDevice(c->device));
    for (int i = 0; i < c->ev_pending; ++i) RRTX_HIP(hipEventSynchronize(c->ev_stop[i]));
    RRTX_HIP(hipStreamSynchronize(c->stream));
    int rc = c->p.fp64 ? upload_scene<double>(c, s) : upload_scene<float>(c, s);
    if (rc) return rc;
    // persistent grid: fill the chip once with the kernel variant this scene selects; never more
    // blocks than there are task batches
    int bpc = 0;
    RRTX_HIP(c->p.fp64 ? render_occupancy<double>(make_params<double>(c, nullptr, c->accel), c->use_filter, c->lds_mode, &bpc) : render_occupancy<float>(make_params<float>(c, nullptr, c->accel), c->use_filter, c->lds_mode, &bpc));
    c->blocks_per_cu = bpc < 1 ? 1 : bpc;
    int64_t grid = (int64_t)c->num_cus * c->blocks_per_cu;
    const int64_t batches = ((int64_t)c->total_tasks + kTaskBatch - 1) / kTaskBatch;
    const int wpb = (!c->accel && c->use_filter && c->lds_mode == 3) ? mf_block_threads(c->fsize) / 64 : kWavesPerBlock; // waves in a block of the render variant this scene selects
    const int64_t need_blocks = (batches + wpb - 1) / wpb;
    if (grid > need_blocks) grid = need_blocks;
    if (grid < 1) grid = 1;
    c->grid_blocks = (int)grid;
    // camera-ray candidate lists (one pre-pass per scene: they depend on the camera and the spheres)
    c->have_plist = c->have_split = c->have_first = false;
    // (a LIST pass tests every moving sphere and triangle per camera ray, with the few lanes that hold one: with a
    // mesh in the scene the walk resp. the scan pass, which has to go through them anyway, is the cheaper way)
    if (!(c->p.flags & RRTX_FLAG_NO_PRIMARY_LISTS) && c->n_sph <= 65535 && c->local_rows > 0 && c->n_tri + c->n_msph <= 64) {
        const size_t bytes = (size_t)c->local_rows * c->p.image_width * kPlistStride * sizeof(uint16_t);
        if (int e = ensure_buffer(c, (void **)&c->d_plist, bytes)) return e;
        uint16_t *pl = c->d_plist;
        c->have_plist = true;
        if (c->p.fp64) {
            KernelParams<double> P = make_params<double>(c, nullptr, c->accel);
            RRTX_HIP(launch_primary_lists<double>(P, pl, c->stream));
        }
        else {
            KernelParams<float> P = make_params<float>(c, nullptr, c->accel);
            RRTX_HIP(launch_primary_lists<float>(P, pl, c->stream));
        }
        RRTX_HIP(hipStreamSynchronize(c->stream));
        // the sky split: scenes of spheres alone (a LIST pass of any other scene tests every moving sphere and triangle too), no taper, a loop that runs at all
        const uint32_t n_px = (uint32_t)((size_t)c->local_rows * c->p.image_width);
        if (c->n_msph == 0 && c->n_tri == 0 && c->taper_pixel == n_px && c->p.max_depth > 0 && c->p.list_passes >= 0 && !(c->p.flags & (RRTX_FLAG_NO_SKY_SPLIT | RRTX_FLAG_VERIFY_LISTS))) {
            const uint32_t blocks = (n_px + 255u) / 256u;
            if (int e = ensure_buffer(c, (void **)&c->d_order, (size_t)n_px * sizeof(uint32_t))) return e;
            if (int e = ensure_buffer(c, (void **)&c->d_order_scratch, ((size_t)blocks + 1) * sizeof(uint32_t))) return e;
            RRTX_HIP(launch_order_pixels(pl, n_px, c->d_order_scratch, c->d_order, c->stream));
            uint32_t n_first = 0;
            RRTX_HIP(hipMemcpyAsync(&n_first, c->d_order_scratch + blocks, sizeof n_first, hipMemcpyDeviceToHost, c->stream));
            RRTX_HIP(hipStreamSynchronize(c->stream));
            if (n_first < n_px) // (a frame without a sky-only pixel keeps the plain queue)
                c->n_queue_pixels = n_first, c->have_split = true;
        }
        // the first bounce of every queued sample as a dense pre-pass of each launch: 8 F per sample (15 GB for 1200x800 spp 500 in fp32) - where the device can
        // spare that and where it pays.  Measured on one box (EXPERIMENTS.md, round 4): with the grid walk 37.3 -> 33.7 ms (fp32) and 60.3 -> 52.6 ms (fp64) at
        // 1200x800 spp 500; by launch size, final.txt with / without: 61 M samples 5.16 / 5.48 ms, 31 M 3.13 / 3.23, 15 M 2.04 / 2.06, 7.7 M 1.43 / 1.42, 3.8 M
        // 1.12 / 1.02 (the same for an eighth of the frame at as many samples); with the list scan nothing anywhere (its loop is bound by the scan passes, which a
        // lane without a camera ray does not shorten).  Otherwise, and for scenes with anything but spheres, the render loop forms its camera rays itself (LIST
        // passes), as up to round 3.
        const bool first_pays = c->accel && ((c->p.flags & RRTX_FLAG_FIRST_BOUNCE_ALWAYS) || (size_t)c->total_tasks * (size_t)c->chunk >= kFirstBounceMinSamples);
        if (first_pays && c->n_msph == 0 && c->n_tri == 0 && c->taper_pixel == n_px && c->p.max_depth > 0 && c->p.list_passes >= 0 && c->n_mat <= 65536 &&
            !(c->p.flags & (RRTX_FLAG_NO_FIRST_BOUNCE | RRTX_FLAG_VERIFY_LISTS))) {
            const size_t bytes = (((size_t)c->total_tasks + 63) / 64) * 64 * (size_t)c->chunk * 8 * c->fsize;
            size_t free_b = 0, total_b = 0;
            if (bytes <= ((size_t)96 << 30) && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                size_t *cap = nullptr;
                for (auto &e : c->caps)
                    if (e.first == &c->d_first) cap = &e.second;
                if (!cap) {
                    c->caps.push_back({&c->d_first, 0});
                    cap = &c->caps.back().second;
                }
                if (!c->d_first || *cap < bytes) { // (exactly what is wanted: no room to grow on a buffer this size)
                    if (c->d_first) (void)hipFree(c->d_first);
                    c->d_first = nullptr, *cap = 0;
                    if (bytes + ((size_t)4 << 30) <= free_b + 0 && hipMalloc(&c->d_first, bytes + 64) == hipSuccess)
                        *cap = bytes;
                    else
                        c->d_first = nullptr, (void)hipGetLastError();
                }
                c->have_first = c->d_first != nullptr;
            }
        }
    }
    // parked-item buffer: every resident wave can park at most kHandoffLanes items
    c->tail_capacity = 0;
    if (c->tail_ok && !(c->p.flags & RRTX_FLAG_NO_TAIL_KERNEL)) {
        const size_t item = c->p.fp64 ? sizeof(TailItem<double>) : sizeof(TailItem<float>);
        c->handoff_lanes = c->p.handoff_lanes > 0 ? (c->p.handoff_lanes > 64 ? 64 : c->p.handoff_lanes) : kHandoffLanes;
        const size_t items = (size_t)c->grid_blocks * wpb * 128; // a wave may park all 64 lanes and up to 64 tasks of its pool
        // (tail_capacity stays 0 - no hand-off - unless all three buffers are there)
        if (int e = ensure_buffer(c, &c->d_tail_items, items * item)) return e;
        if (int e = ensure_buffer(c, &c->d_tail_rad, items * kTailSplit * 3 * c->fsize)) return e;
        if (int e = ensure_buffer(c, (void **)&c->d_tail_units, items * kTailSplit * sizeof(uint32_t))) return e;
        c->tail_capacity = items;
        int64_t tb = (int64_t)(c->tail_capacity + kWavesPerBlock - 1) / kWavesPerBlock; // one wave per item at most
        const int64_t cap = (int64_t)c->num_cus * 8;
        c->tail_blocks = (int)(tb < cap ? tb : cap);
        if (c->tail_blocks < 1) c->tail_blocks = 1;
        c->resume_blocks = (c->grid_blocks * wpb + kWavesPerBlock - 1) / kWavesPerBlock; // (as many waves as the render pass had; persistent waves pulling units: blocks beyond what is resident just find the queue empty)
#ifdef RRTX_EXPERIMENTS
        c->resume_blocks = (int)grid_knob("RRTX_RESUME_BLOCKS", (double)c->resume_blocks);
#endif
    }
    return RRTX_OK;
}

int rrtx_shard_rows(const rrtx_ctx *c, int32_t *rows, int cap)
{
    if (!c) return fail(RRTX_E_INVALID, "rrtx_shard_rows: null context");
    std::vector<int32_t> r;
    int n = rows_of_shard(c->p, &r);
    if (rows)
        for (int i = 0; i < n && i < cap; ++i) rows[i] = r[i];
    return n;
}

int rrtx_render_device(rrtx_ctx *c, void *d_rows, void *hip_stream)
{
    if (!c || !d_rows) return fail(RRTX_E_INVALID, "rrtx_render_device: null argument");
    if (!c->have_scene) return fail(RRTX_E_NO_SCENE, "rrtx_render_device: no scene set");
    RRTX_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)hip_stream; // NULL is HIP's null stream, as everywhere in the HIP API
    if (c->ev_pending == kEventRing) {
        int rc = drain_events(c, nullptr);
        if (rc) return rc;
    }
    if (c->total_tasks == 0) return RRTX_OK;
    // task cursor, parked-item count, tail cursor, unit count, queue-over flag (256 bytes) - and, behind them, the counters
#if defined(RRTX_SECTION_DIAG) || defined(RRTX_RESUME_DIAG)
    RRTX_HIP(hipMemsetAsync(c->d_queue, 0, 512, st));
#else
    RRTX_HIP(hipMemsetAsync(c->d_queue, 0, 256 + 64, st)); // (counters 0 .. 7: statistics, and the convergence faults, which are counted whether statistics are on or not)
#endif
    const int slot = c->ev_pending;
    RRTX_HIP(hipEventRecord(c->ev_start[slot], st));
    void *out = c->use_partial ? c->d_partial : d_rows;
    const uint32_t n_px = (uint32_t)((size_t)c->local_rows * c->p.image_width);
    const FinalizeShape shape = {n_px, c->taper_pixel, c->chunks_per_pixel, c->chunk, c->p.samples_per_pixel, 0, c->per_sample ? 1 : 0};
    // The launches of one render.  If one of them fails after an earlier one was enqueued, that kernel may still be reading the
    // scene's tables while the caller, holding an error, goes on to rrtx_set_scene (which waits for stop EVENTS only, and none
    // will have been recorded): the stream is drained before the error is returned.
    auto enqueue = [&](auto fp) -> hipError_t {
        typedef decltype(fp) F;
        KernelParams<F> P = make_params<F>(c, out, c->accel);
        hipError_t e = hipSuccess;
        bool forked = false;
        // (the first bounce first: the sky kernel is to run beside the persistent render kernel - beside another dense kernel it would only share the device with it)
        if (c->have_first) e = launch_first_bounce<F>(P, P.total_tasks, c->num_cus, st); // (the queue's positions: every task, or those of the pixels the sky split left)
        // the tasks of the sky-only pixels: a dense kernel of camera rays (rrtx_kernels.hip: the sky split).  Beside the render kernel: its stream waits for what
        // precedes the render kernel on the caller's stream (this launch's fill of the control words), the caller's stream waits for it before the sums are formed;
        // and it is enqueued after the render kernel.  (Measured, kernel trace: the device starts it 0.13 ms BEFORE the render kernel either way, and its 0.42 ms
        // disappear in the render kernel's first half millisecond, where 390 000 lanes all fetch tables, pull tasks and form camera rays at once: C3 50.24 -> 49.77 ms,
        // use_bvh 36.56 -> 35.88, C4 use_bvh 56.1 -> 55.6.)
        if (e == hipSuccess && c->have_split && !(c->p.flags & RRTX_FLAG_SKY_SAME_STREAM)) {
            e = hipEventRecord(c->ev_fork, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->side_stream, c->ev_fork, 0);
            forked = e == hipSuccess;
        }
        auto launch_sky = [&]() {
            const uint32_t first = c->n_queue_pixels * (uint32_t)c->chunks_per_pixel;
            hipError_t es = launch_sky_tasks<F>(P, first, c->total_tasks - first, c->num_cus, forked ? c->side_stream : st);
            if (forked) {
                const hipError_t e2 = hipEventRecord(c->ev_join, c->side_stream);
                if (es == hipSuccess) es = e2;
            }
            return es;
        };
        const bool sky_first = !forked;
        if (e == hipSuccess && c->have_split && sky_first) e = launch_sky();
        if (e == hipSuccess) e = launch_render<F>(P, c->use_filter, c->lds_mode, c->grid_blocks, st);
        if (e == hipSuccess && c->have_split && !sky_first) e = launch_sky();
        if (e == hipSuccess && c->tail_capacity)
            e = (c->accel || c->tail_grid) ? launch_resume<F>(make_params<F>(c, out, true), c->use_filter, c->resume_blocks, st) : launch_tail<F>(P, c->use_filter, c->tail_blocks, st);
        if (forked) { // (also after an error: whatever went to the side stream is waited for by the caller's stream, which the error path drains)
            const hipError_t e2 = hipStreamWaitEvent(st, c->ev_join, 0);
            if (e == hipSuccess) e = e2;
        }
        if (e == hipSuccess && c->use_partial) e = launch_finalize<F>((const F *)c->d_partial, (F *)d_rows, shape, st);
        if (e == hipSuccess) e = hipEventRecord(c->ev_stop[slot], st);
        return e;
    };
    const hipError_t le = c->p.fp64 ? enqueue(double()) : enqueue(float());
    if (le != hipSuccess) {
        (void)hipStreamSynchronize(st);
        char buf[256];
        snprintf(buf, sizeof buf, "rrtx_render_device: HIP error = %u (%s) while enqueueing the render", (unsigned)le, hipGetErrorString(le));
        return fail(RRTX_E_DEVICE, buf);
    }
    c->ev_pending = slot + 1;
    return RRTX_OK;
}

void *rrtx_stream(rrtx_ctx *c) { return c ? (void *)c->stream : nullptr; }

#ifdef RRTX_DIAG
extern "C" int rrtx_diag_read(rrtx_ctx *c, void *dst)
{
    RRTX_HIP(hipDeviceSynchronize());
    RRTX_HIP(hipMemcpy(dst, c->d_diag, (size_t)131072 * 64, hipMemcpyDeviceToHost));
    RRTX_HIP(hipMemset(c->d_diag, 0, (size_t)131072 * 64));
    return 0;
}
#endif

#ifdef RRTX_RESUME_DIAG
extern "C" int rrtx_resume_diag(rrtx_ctx *c, unsigned long long out[8]) // developer builds: the resume pass's longest wave (iterations, cycles); iterations of all waves, waves with work, segments
{
    RRTX_HIP(hipDeviceSynchronize());
    RRTX_HIP(hipMemcpy(out, c->d_counters + 24, 64, hipMemcpyDeviceToHost));
    return 0;
}
#endif
#ifdef RRTX_SECTION_DIAG
extern "C" int rrtx_dense_diag(rrtx_ctx *c, unsigned long long out[8]) // developer builds: what the dense pairing of the accelerated variants was fed (rrtx_kernels.hip, dense_dbg)
{
    RRTX_HIP(hipDeviceSynchronize());
    RRTX_HIP(hipMemcpy(out, c->d_counters + 8, 64, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int rrtx_section_diag(rrtx_ctx *c, unsigned long long out[8]) // developer builds: a wave's clock cycles per section of the render loop, summed over the waves
{
    RRTX_HIP(hipDeviceSynchronize());
    RRTX_HIP(hipMemcpy(out, c->d_counters + 16, 64, hipMemcpyDeviceToHost));
    return 0;
}
#endif

int rrtx_collect(rrtx_ctx *c, rrtx_stats *stats)
{
    if (!c) return fail(RRTX_E_INVALID, "rrtx_collect: null context");
    RRTX_HIP(hipSetDevice(c->device));
    double last_ms = 0.0;
    const double before_total = c->kernel_ms_total;
    const long before_n = c->renders_total;
    int rc = drain_events(c, &last_ms);
    if (rc) return rc;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->kernel_ms = last_ms;
        stats->kernel_ms_sum = c->kernel_ms_total - before_total;
        stats->renders = (int32_t)(c->renders_total - before_n);
        stats->wall_ms = c->last_wall_ms;
        stats->samples = (uint64_t)c->local_rows * c->p.image_width * (uint64_t)c->p.samples_per_pixel;
        stats->sky_pixels = c->have_split ? (int32_t)((size_t)c->local_rows * c->p.image_width - c->n_queue_pixels) : 0;
        stats->first_bounce = c->have_first ? 1 : 0;
        {
            unsigned long long faults = 0;
            RRTX_HIP(hipMemcpy(&faults, c->d_counters + 6, sizeof faults, hipMemcpyDeviceToHost));
            stats->convergence_faults = faults;
            // (a frame rendered by a wave that was short of lanes in a wave-wide step is not to be trusted: an error, not a statistic)
            if (faults) return fail(RRTX_E_DEVICE, "rrtx_collect: " + std::to_string(faults) + " wave(s) reached a wave-wide step (matrix-core scan / dense pairing) with lanes masked off; the frame is invalid");
        }
        if (c->p.collect_stats) {
            unsigned long long ctr[6] = {0, 0, 0, 0, 0, 0};
            RRTX_HIP(hipMemcpy(ctr, c->d_counters, sizeof ctr, hipMemcpyDeviceToHost));
            stats->list_mismatches = (int32_t)(ctr[2] > 0x7fffffffull ? 0x7fffffffull : ctr[2]);
            const unsigned long long seg = ctr[0];
            stats->segments = seg;
            stats->candidates = ctr[1];
            stats->scanned_segments = ctr[3];
            stats->walk_cells = ctr[4], stats->walk_pairs = ctr[5];
            const uint64_t nprim = (uint64_t)c->n_sph + c->n_msph + c->n_tri;
            stats->prim_tests = seg * nprim;
            // B_prim = 4 scalars (sphere) / 9 scalars (moving sphere, triangle)
            stats->bytes_algorithmic = seg * ((uint64_t)c->n_sph * 4 + (uint64_t)(c->n_msph + c->n_tri) * 9) * c->fsize +
                                       (uint64_t)c->local_rows * c->p.image_width * 3 * c->fsize;
        }
        stats->grid_blocks = c->grid_blocks;
        stats->block_threads = (!c->accel && c->use_filter && c->lds_mode == 3) ? mf_block_threads(c->fsize) : kBlockThreads;
        stats->sample_chunk = c->sum_chunk; // (the summation shape; per-sample launches schedule in tasks of c->chunk)
        stats->accel_cells = c->accel ? c->n_grid_cells : 0;
        stats->accel_exact = c->accel ? (c->accel_exact ? 1 : 0) : 1;
        stats->local_rows = c->local_rows;
        stats->scan_filter = c->use_filter ? 1 : 0;
        stats->scan_mfma = (!c->accel && c->use_filter && c->lds_mode == 3) ? 1 : 0;
    }
    return RRTX_OK;
}

int rrtx_render(rrtx_ctx *c, void *fb, rrtx_stats *stats)
{
    if (!c || !fb) return fail(RRTX_E_INVALID, "rrtx_render: null argument");
    if (!c->have_scene) return fail(RRTX_E_NO_SCENE, "rrtx_render: no scene set");
    RRTX_HIP(hipSetDevice(c->device));
    const size_t row_bytes = (size_t)c->p.image_width * 3 * c->fsize;
    const size_t bytes = row_bytes * (size_t)c->local_rows;
    if (!c->d_rows && bytes) RRTX_HIP(hipMalloc(&c->d_rows, bytes));
    auto t0 = std::chrono::steady_clock::now();
    if (bytes) {
        int rc = rrtx_render_device(c, c->d_rows, (void *)c->stream);
        if (rc) return rc;
    }
    RRTX_HIP(hipStreamSynchronize(c->stream));
    auto t1 = std::chrono::steady_clock::now();
    c->last_wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (bytes) {
        // straight into the caller's frame (DMA at link speed when it is page-locked, rrtx_pin_host): the
        // whole block when the shard is the frame, else run by run of consecutive rows (a tile, or several
        // tiles that happen to touch)
        std::vector<int32_t> rows;
        rows_of_shard(c->p, &rows);
        for (size_t k = 0; k < rows.size();) {
            size_t e = k + 1;
            while (e < rows.size() && rows[e] == rows[e - 1] + 1) ++e;
            RRTX_HIP(hipMemcpyAsync((unsigned char *)fb + (size_t)rows[k] * row_bytes, (const unsigned char *)c->d_rows + k * row_bytes, (e - k) * row_bytes, hipMemcpyDeviceToHost, c->stream));
            k = e;
        }
        RRTX_HIP(hipStreamSynchronize(c->stream));
    }
    return rrtx_collect(c, stats);
}

} // extern "C"
