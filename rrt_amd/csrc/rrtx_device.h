// Device-side data layout of the render path (shared by the packer in rrtx_api.cpp and the
// kernels in rrtx_kernels.hip).  See DESIGN.md "Data layout in HBM".
//
// The reference keeps every primitive as a heap object behind a vtable (rrt.cu:124-174); here the
// world is a handful of flat, read-only arrays:
//   * sphere "hot" records  {cx, cy, cz, r*r}            16 B (fp32) / 32 B (fp64)   — the only bytes
//     the brute-force scan touches per test; read with wave-uniform scalar loads (4 SGPRs)
//   * sphere "cold" records {radius, material}           — read once per hit
//   * moving-sphere / triangle records with everything that sphere.h / moving_sphere.h /
//     triangle.h recompute per call but that depends on the primitive only (c1-c0, t1-t0, r*r,
//     edge1, edge2, face normal) hoisted to the host — same IEEE operation, same operands, so
//     bit-identical to evaluating it per ray
//   * material records {albedo rgb, fuzz|ir, type}
#ifndef RRTX_DEVICE_H
#define RRTX_DEVICE_H

#include <stdint.h>

namespace rrtx {

constexpr int kSphereUnroll = 8;  // tests per straight-line block (fp32; fp64 uses half)
constexpr int kSpherePad = 16;    // the sphere tables are padded to a multiple of this (two blocks)
constexpr int kLdsSceneBytes = 48 * 1024; // largest scan table mirrored in LDS
constexpr int kLdsMfBytes = 36 * 1024;    // ... and largest table of f16 operands for the filter on the matrix cores (64 bytes per sphere, whole blocks of 32: one copy per block of 512 / 768 threads, two such blocks / one to a CU)
#ifndef RRTX_CAND_CAP
#define RRTX_CAND_CAP 16
#endif
constexpr int kCandCap = RRTX_CAND_CAP;      // candidate slots per lane (LDS), flushed when nearly full
#ifndef RRTX_BLOCK_THREADS
#define RRTX_BLOCK_THREADS 256
#endif
constexpr int kBlockThreads = RRTX_BLOCK_THREADS;
constexpr int kWavesPerBlock = kBlockThreads / 64;
// ... of the list-scan variant whose filter runs on the matrix cores: its table of f16 operands (64 bytes per sphere, one copy per
// block in LDS) is shared by twice the waves
#ifndef RRTX_MF_BLOCK_THREADS
#define RRTX_MF_BLOCK_THREADS 512
#endif
#ifndef RRTX_MF_BLOCK_THREADS_F64
#define RRTX_MF_BLOCK_THREADS_F64 768 // (fp64 ray records and result slots: 6.75 KB a wave - ONE block of twelve waves to a CU, three to a SIMD: 66.5 ms; 256 threads, two blocks = 2 waves per SIMD: 79.2; 384, six waves spread unevenly over the four SIMDs: 104.8)
#endif
constexpr int mf_block_threads(size_t fsize) { return fsize == 4 ? RRTX_MF_BLOCK_THREADS : RRTX_MF_BLOCK_THREADS_F64; }
// A first-bounce record: part 0 = {o.x, o.y, o.z, d.x}, part 1 = {d.y, d.z, code, -}; code (an integer of F's size, its low 32 bits used): kind << 30 | draws << 16 | material.
//   kFirstRay     : the path goes on from o along d at depth 1; its attenuation is the material's albedo (1 for a dielectric), its generator stands at draw `draws`
//   kFirstDone    : the path ended at its first segment - sky, absorbed, or a depth limit of 1 -, o is its radiance
//   kFirstUnknown : not resolved here (a pixel whose candidate list overflowed; more than 16 383 draws): the render kernel traces the sample from its camera ray
// Layout: part p of sample k of a task at [((task / 64) x chunk + k) x 2 + p][task % 64], 4 F each - the dense kernel runs one lane per task, so the 64 lanes of a wave
// write a part side by side (1 KB per store instruction in fp32), and the render kernel's lanes, which hold neighbouring tasks, read them the same way.
constexpr uint32_t kFirstRay = 0u, kFirstDone = 1u, kFirstUnknown = 3u;
template <typename F> struct FirstCode;
template <> struct FirstCode<float> {
    typedef uint32_t type;
};
template <> struct FirstCode<double> {
    typedef uint64_t type;
};
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    size_t first_slot(uint32_t task, uint32_t k, uint32_t chunk, uint32_t part)
{
    return (((size_t)(task >> 6) * chunk + k) * 2u + part) * 64u + (task & 63u);
}

constexpr uint32_t kTaskBatch = 64;    // chunk tasks a wave pulls from the global queue at a time, at most ...
constexpr uint32_t kTaskBatchMin = 8; // ... and at least (guided: batches shrink towards the end of the chunk tasks)
constexpr uint32_t kQueueOverFlag = 32; // P.queue[32] (its own 128-byte line): set once the cursor has passed the last task
constexpr uint32_t kTaperBatch = 256; // single-sample tasks per pull, at most (at least kTaskBatch)
// Safety factor of the conservative scan filter, in units of the unit roundoff.  The
// analytic bound needs about 160; empirically false negatives appear only below 16.
constexpr int kFilterK = 256;
// The filter is ALWAYS evaluated in fp32, also for fp64 rays (their fp64 FMAs run at half rate): ray and
// centres are rounded to float, and the margin covers that rounding too.  Empirically false negatives
// against the fp64 discriminant vanish at 16 here as well (tests/test_filter_bound.py).
constexpr int kFilterK64 = 512;
// ... and of its form on the matrix cores (f16 x 2 operands, rrtx_pack.h: pack_mf_table), twice the margin each
constexpr int kFilterKMf = 512, kFilterKMf64 = 1024;
// A wave of the render kernel hands its unfinished items to the tail kernel once the queue is dry and
// at most this many of its lanes are alive (break-even of one-ray-per-wave against one-ray-per-lane).
constexpr int kHandoffLanes = 7;
constexpr int kPlistCap = 15;    // sphere indices per pixel in the camera-ray candidate lists ...
constexpr int kPlistStride = 16; // ... stored as uint16 [count | 0xFFFF, idx...]: 32 bytes per pixel
constexpr int kListPasses = 3;   // LIST passes allowed between two SCAN passes (measured: 1 -> 80.9, 2 -> 79.8, 3 -> 79.0, 6 -> 78.7 ms)
constexpr int kTailSplit = 8;    // a parked item is finished as up to this many independent units (3 bits in tail_units)
// ... and after this many iterations past queue-dry regardless of the lane count.  Round 4, with the resume pass's units dealt (a pass costs 60 us less than it did): the
// list scan and the variants for scenes of spheres alone 4 / 8 / 12 / 16 / 20 / 24: C2 1.115 / 1.067 / 1.035 / 0.989 / 0.975 / 0.982 ms, test3.txt spp 10 1.51 / 1.42 / 1.31 /
// 1.28 / 1.26 / 1.28, an eighth of C3 6.83 / 6.80 / 6.78 / 6.76 / 6.79 / 6.75, the whole of it flat; the densely pairing variants (a mesh under use_bvh: an iteration of
// theirs is 16 us, and the resume pass repacks what is left) 2 / 4 / 8 / 12 / 20: 5.16 / 5.15 / 5.30 / 5.47 / 5.73 ms, fp64 6.05 / 6.04 / 6.22 / 6.49 / 7.10.
constexpr int kHandoffIters = 20, kHandoffItersDense = 4;

template <typename F> struct alignas(4 * sizeof(F)) SphereHot {
    F cx, cy, cz, r2;
};
template <typename F> struct alignas(2 * sizeof(F)) SphereCold {
    F radius;
    int32_t mat;
};
template <typename F> struct MovingSphereRec {
    F c0[3];
    F dc[3];  // center1 - center0            (moving_sphere.h:29)
    F t0, dt; // time0, time1 - time0         (moving_sphere.h:29)
    F r2;     // radius * radius              (moving_sphere.h:38)
    F radius;
    int32_t mat;
};
template <typename F> struct TriangleRec {
    F v0[3];
    F e1[3]; // vertices[1] - vertices[0]     (triangle.h:39)
    F e2[3]; // vertices[2] - vertices[0]     (triangle.h:40)
    F n[3];  // get_normal()                  (triangle.h:9-15)
    int32_t mat;
};
// What the scan's triangle phase reads per triangle: the nine numbers of the test in ONE aligned
// s_load_dwordx16 (fp64: two) instead of four scalar loads of the 52-byte TriangleRec's fields (a mesh of
// 27 k triangles: 502 -> 486 ms; dropping the camera-ray LIST passes, which walk all triangles with a
// quarter of the lanes, brought the larger step: 486 -> 363 ms).
template <typename F> struct alignas(16 * sizeof(F)) TriScanRec {
    F v0[3], e1[3], e2[3], pad[7];
};
template <typename F> struct alignas(4 * sizeof(F)) MaterialRec {
    F r, g, b;
    F param; // metal: min(fuzz,1) (material.h:48) | dielectric: ir
    int32_t type;
    int32_t pad[3];
};
template <typename F> struct CameraRec { // camera.h:43-48
    F origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    F lens_radius, time0, time1;
};

// A parked work item (render kernel -> tail kernel)
template <typename F> struct TailItem {
    uint32_t task;
    int32_t s_cur, depth;
    uint32_t need_ray;
    uint32_t k0, k1, n, pad; // RNG stream of the current sample and its draw counter
    F acc[3];                // partial sum of the task so far
    F o[3], d[3], tm, atten[3];
};

// Acceleration grid (use_bvh != 0; DESIGN.md "Accelerated closest hit").  A uniform grid over the
// boxes of all primitives that are not much larger than a cell; the few that are (the ground sphere)
// sit in an "always" list tested for every segment.  Cell c lists primitives cell_start[c] ..
// cell_start[c + 1] of cell_prims (unified primitive indices).
typedef uint32_t GridPrim; // an entry of cell_prims (16 bits were enough for final.txt; a mesh of 100 000 triangles is not)
template <typename F> struct GridRec {
    F gmin[3], gmax[3];  // box of the grid
    F cell[3], inv_cell[3];
    int32_t dims[3];
    F center[3];
    F far2;              // rays starting further than sqrt(far2) from `center` take the list scan (their
                         // exact tests are too inaccurate for the inflation the cells were built with)
    F slack, slack1, half_diag; // the walk continues slack + slack1 (|o - center| + half_diag) world units beyond the closest hit so far
    int32_t max_steps;   // bound on the trips of the walk (cells stepped through + primitives tested)
    F dir2_max;          // rays with |d|^2 above this take the list scan: the inflation of gridded triangles is proven up to it
    int32_t walk_slice;  // cells a lane walks per iteration of the render loop before the others get their turn (4 .. 16)
    // Empty-space skipping (large, mostly empty grids: meshes): one byte per block of 4 x 4 x 4 cells - does any of them hold an entry? -
    // stored behind cell_start's cells + 1 words, from word coarse_off on (0: none).  The batched walk (accel_walk_prepare) crosses an empty
    // block in ONE step.
    int32_t coarse_dims[3];
    int32_t coarse_off;
};

// Division by a launch constant: n / d == umulhi(n, m) >> shift for every n < 2^31 (m = floor(2^(31 + L) / d) + 1,
// L = ceil(log2 d), shift = L - 1; Granlund & Montgomery).  A runtime udiv costs ~40 VALU instructions
// and the task decoding needs several per camera ray.
struct FastDiv {
    uint32_t m, shift, is_one, d;
};
inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f = {0u, 0u, d <= 1u ? 1u : 0u, d};
    if (d > 1u) {
        uint32_t L = 0;
        while ((1ull << L) < d) ++L;
        f.m = (uint32_t)(((unsigned __int128)1 << (31 + L)) / d + 1);
        f.shift = L - 1;
    }
    return f;
}

template <typename F> struct KernelParams {
    const SphereHot<F> *sph_hot;    // n_sph_padded records {cx, cy, cz, r*r}: the exact test
    const SphereHot<float> *sph_filter; // n_sph_padded records {cx, cy, cz, thr}: the conservative scan filter (fp32 for every F)
    const uint32_t *mf_table;           // LDSMODE = 3: the filter's sphere operands for the matrix cores, n_sph_padded x 64 bytes (rrtx_pack.h, pack_mf_table)
    const uint32_t *mf_big;             // ... and the spheres that table cannot hold (n_mf_big of them, tested exactly)
    int32_t n_mf_big;
    const SphereCold<F> *sph_cold;
    const MovingSphereRec<F> *msph;
    const TriangleRec<F> *tri;
    const TriScanRec<F> *tri_scan; // the same triangles, as the scan reads them
    const MaterialRec<F> *mat;
    int32_t n_sph, n_sph_padded, n_msph, n_tri;
    CameraRec<F> cam;
    int32_t W, H, spp, max_depth;
    uint32_t seed;
    int32_t chunk;           // samples per task
    int32_t chunks_per_pixel;
    int32_t per_sample;      // 1: a task does not add its samples up - every sample's radiance goes to out[(pixel * spp + sample) * 3] and finalize_kernel adds a
                             // pixel's samples in the ORDER ASKED FOR (the reference's: one running sum, rrt.cu:115), whatever the tasks' size.  That is how
                             // sample_chunk = -1 runs at the speed of 8-sample work items instead of one 500-sample item per pixel.
    // shard: local row lr -> global row ((lr / tile_rows) * shard_count + shard_rank) * tile_rows + lr % tile_rows
    int32_t local_rows, tile_rows, shard_rank, shard_count;
    FastDiv div_cpp, div_spp, div_w, div_tile; // chunks_per_pixel, spp, W, tile_rows
    uint32_t taper_pixel;    // local pixels from this one on are cut into single-sample tasks (== pixel count: none)
    uint32_t taper_task_base; // taper_pixel * chunks_per_pixel: index of the first single-sample task
    uint32_t total_tasks;    // taper_task_base + (local_rows * W - taper_pixel) * spp
    uint32_t *queue;         // [0] global task cursor, [kQueueOverFlag] "queue over" flag (zeroed before every launch)
    F *out;                  // [total_tasks][3]: per-task partial sums, see task_slot (== the local frame when every pixel is one task)
    unsigned long long *counters; // [0] segments, [1] candidates refined in phase 2 (only if collect_stats)
    int32_t collect_stats;
    int32_t handoff_lanes;   // 0 = never hand off (the render kernel finishes everything itself)
    int32_t handoff_iters;   // ... and unconditionally this many iterations after the queue ran dry
    uint32_t *tail_count;    // [0] parked items, [1] the tail kernel's unit cursor, [2] units (zeroed before every launch)
    TailItem<F> *tail_items; // capacity: resident waves * 64
    uint32_t *tail_units;    // work units of the parked items: item << 3 | unit
    F *tail_rad;             // [item][kTailSplit][3] results of the units
    unsigned long long *diag; // RRTX_DIAG builds only (timing stamps), otherwise unused
    // accelerated closest hit (all null / 0 when the list scan is used)
    const uint32_t *grid_cell_start; // [cells + 1]
    const GridPrim *grid_cell_prims;
    const uint32_t *grid_always;
    int32_t n_always, n_grid_cells, n_grid_prims;
    GridRec<F> grid;
    const uint16_t *plist;   // camera-ray candidate lists [local pixel][kPlistStride], or nullptr
    // The sky split (scenes of spheres alone with candidate lists): a pixel whose list is EMPTY is sky in every sample - its tasks are eight one-segment paths each, finished
    // by sky_tasks_kernel, a dense kernel of nothing but camera rays, and never queued.  pixel_order is a stable partition of the local pixels, those with a non-empty list
    // (or an overflowed one) first, n_queue_pixels of them: the render kernel's queue is positions [0, n_queue_pixels x chunks_per_pixel), the sky kernel takes the rest.
    const uint32_t *pixel_order; // position -> local pixel, or nullptr: no split, the queue is every task as it lies
    uint32_t n_queue_pixels;
    // The first bounce, done densely before the render kernel starts (first_bounce_kernel; scenes of spheres alone with candidate lists): per sample of every queued
    // task ONE record of 8 F - see FirstCode, first_slot - that says how the sample's camera ray fared: the path ended (its radiance), or it goes on (the scattered
    // ray, the material whose albedo is its attenuation, the draws consumed).  The render kernel then forms no camera ray, runs no LIST pass and shades no primary hit.
    const void *first;           // [first_slot(task, sample of its chunk, part)] of 4 F each, or nullptr
    int32_t list_passes;     // 0 = every segment goes through the scan
    int32_t verify_lists;    // test mode: counters[2] counts camera rays whose list hit differs from the full scan
};

// shape of the partial-sum buffer, for finalize_kernel
struct FinalizeShape {
    uint32_t n_pixels, taper_pixel; // local pixels; first single-sample pixel
    int32_t chunks_per_pixel, chunk, spp;
    int32_t group;                  // pixels a block stages at a time (their slabs fit the kernel's LDS)
    int32_t per_sample;             // 1: `partial` holds every sample, [pixel][sample][3]; the frame is one running sum per pixel (finalize_samples_kernel)
};

// shape of a multi-device gather, for deinterleave_kernel (rrtx_group.cpp)
constexpr int kMaxGroup = 64;
struct GatherShape {
    uint32_t row_values; // W * 3
    uint32_t height, tile_rows, n_shards;
    uint32_t row_off[kMaxGroup]; // rows of the shards before shard r in the gathered buffer
};

// The row-tile plan, in ONE place for the three that must agree: which frame rows a shard renders (rrtx_api.cpp: rows_of_shard,
// rrtx_shard_rows), where a shard's local row lies in the frame (the render kernel: task_decode), and where a frame row lies in the
// gathered buffer (deinterleave_kernel).  tests/gather_plan_check.cpp compiles these for the host and holds them against each other
// and against rrt_amd/dist.py's shard_rows for N = 1 .. 9 members, ragged last tiles and more members than tiles.
#if defined(__HIPCC__)
#define RRTX_HD __host__ __device__ __forceinline__
#else
#define RRTX_HD inline
#endif
// rows of an H-row frame that belong to shard `rank` of N (tiles of T rows, tile t to shard t mod N)
RRTX_HD uint32_t shard_row_count(uint32_t H, uint32_t T, uint32_t N, uint32_t rank)
{
    const uint32_t tiles = (H + T - 1u) / T;            // the last one may be ragged
    const uint32_t mine = tiles > rank ? (tiles - rank + N - 1u) / N : 0u; // tiles rank, rank + N, ...
    if (mine == 0u) return 0u;
    const uint32_t last = rank + (mine - 1u) * N;       // my last tile: ragged if it is the frame's last
    return mine * T - (last == tiles - 1u ? tiles * T - H : 0u);
}
// local row lr of shard `rank` -> frame row (what task_decode computes with its launch-constant divisions)
RRTX_HD uint32_t shard_local_to_frame_row(uint32_t lr, uint32_t T, uint32_t N, uint32_t rank)
{
    const uint32_t tile = lr / T;
    return (tile * N + rank) * T + (lr - tile * T);
}
// frame row j -> row of the gathered buffer (the shards' compact blocks side by side, shard r from row_off[r] on)
RRTX_HD uint32_t gather_source_row(const GatherShape &S, uint32_t j)
{
    const uint32_t tile = j / S.tile_rows, r = tile % S.n_shards;
    return S.row_off[r] + (tile / S.n_shards) * S.tile_rows + (j - tile * S.tile_rows);
}

// Samples handed out as single-sample tasks at the end of the queue, per compute unit, when
// rrtx_params.taper_samples is 0 (automatic).  Measured on final.txt 1200x800 with the queue-over flag,
// pool parking and the unit split of parked items in place: none is best from spp 48 up (spp 504: 78.6 ms
// without, 79.6 ms with 2048 * 24 per CU: the cursor becomes the bottleneck in cheap regions of the
// frame); only very short launches gain (spp 8: 2.9 vs 3.2 ms).  So: off unless asked for.
constexpr int64_t kTaperSamplesPerCu = 0;

} // namespace rrtx

#endif
