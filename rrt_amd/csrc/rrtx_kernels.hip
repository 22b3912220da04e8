// Render megakernel for gfx950 (MI355X).  Replaces render_init + cuda_render + ray_color and all
// the virtual hit()/scatter() calls behind them (rrt.cu:42-122, hittable_list.h:95-117,
// sphere.h:33-58, moving_sphere.h:27-58, triangle.h:35-75, material.h:21-109, camera.h:31-38).
//
// Shape of the computation (DESIGN.md "Kernel"):
//   * persistent waves; a work item ("task") is (pixel, chunk of consecutive samples).  Waves pull
//     batches of tasks from one global cursor and hand them to lanes with a wave64 ballot +
//     prefix-popcount, so a lane whose task is finished starts the next one in the same loop
//     iteration — every live lane always carries a ray into the primitive scan.
//   * the scan is the reference's linear hittable_list scan.  Phase 1 walks the sphere table with
//     WAVE-UNIFORM SCALAR LOADS (16 B/sphere into SGPRs, no VGPR or LDS traffic) and evaluates only
//     the discriminant; lanes whose discriminant is not negative append the primitive index to a
//     per-lane candidate list in LDS.  Phase 2 lets every lane walk ITS OWN short list and run the
//     exact root / range / tie logic of the reference, in list (= primitive) order, so the result is
//     identical to the sequential scan with its shrinking t_max, including "a later sphere wins an
//     exact tie" (sphere.h:46-48) and the strict range of triangles (triangle.h:63).
//   * no FMA contraction anywhere (-ffp-contract=off): rrtc's fp32 image depends on unfused
//     rounding, most visibly in c = |oc|^2 - r^2 on the r = 1000 ground sphere.
//   * counter-based RNG keyed by (seed, global pixel index, sample) — no state in memory, no
//     render_init, identical image for any sharding of the frame.
//
// Matrix cores: ONE part of the path has a matrix form - the conservative filter of the list scan, a dot product of 31 f16 terms
// per (ray, sphere) - and runs as two chained v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays for scenes of spheres alone
// (LDSMODE = 3).  Everything else is per-lane IEEE / integer work on the vector unit.
#include <hip/hip_runtime.h>

#include "rrtx_device.h"
#include "rrtx_path.h" // the arithmetic of the path (shared with the host-side check)
#include "rrtx_wave.h" // wave64 prefix scans (DPP)

namespace rrtx {

// ---------------------------------------------------------------------------------------------
// Candidate push.  `test < thr` false (or unordered) => append primitive index `k0 + u` to this
// lane's list.  Hand-scheduled because the compiler's lowering of "if (cand) push" costs three
// scalar instructions per test (s_and_saveexec / s_cbranch_execz / s_or exec) on the path where NO
// lane is a candidate — 78 % of the tests on final.txt — and once the test itself is down to 8 VALU
// instructions those scalar slots limit the wave.  Here that path is ONE scalar branch.
//   v_cmp_ngt  vcc = !(thr > value)            (true for NaN, like the reference's !(disc < 0))
//   s_cbranch_vccz skip                        wave-uniform: nobody is a candidate
//   s_and_saveexec  /  ds_write_b32 list[cnt] = k0 + u ; cnt += 1  /  restore exec
// LDS ops of one wave execute in order, so the later ds_read of the drain sees these writes; the
// extra lgkmcnt they hold only makes compiler-placed waits more conservative.
// ---------------------------------------------------------------------------------------------
#ifndef RRTX_ASM_PUSH
#define RRTX_ASM_PUSH 1
#endif
#ifndef RRTX_LDS_BLOCK
#define RRTX_LDS_BLOCK 4 // records per LDS-sourced block: 16 VGPRs of operands keep the kernel at 6 waves/SIMD
#endif

#define RRTX_PUSH_ASM(CMP, THRC)                                                                                       \
    asm volatile(CMP " vcc, %[thr], %[val]\n\t"                                                                        \
                     "s_cbranch_vccz 1f\n\t"                                                                           \
                     "s_and_saveexec_b64 %[save], vcc\n\t"                                                             \
                     "v_lshl_add_u32 %[addr], %[cnt], 8, %[base]\n\t"                                                  \
                     "v_mov_b32 %[tmp], %[k0]\n\t"                                                                     \
                     "v_add_u32 %[tmp], %[u], %[tmp]\n\t"                                                              \
                     "ds_write_b32 %[addr], %[tmp]\n\t"                                                                \
                     "v_add_u32 %[cnt], 1, %[cnt]\n\t"                                                                 \
                     "s_mov_b64 exec, %[save]\n"                                                                        \
                     "1:"                                                                                               \
                 : [cnt] "+v"(cnt), [addr] "=&v"(addr), [tmp] "=&v"(tmp), [save] "=&s"(save)                            \
                 : [thr] THRC(thr), [val] "v"(value), [base] "v"(lane_lds_addr), [k0] "s"(k0), [u] "n"(U)               \
                 : "vcc", "scc", "memory")

// lane_lds_addr = LDS byte address of this lane's slot 0; slot s is 256 bytes further (64 lanes x 4 B).
// THR_IN_VGPR: the threshold came from LDS (vector register) instead of a scalar load.
template <int U, bool THR_IN_VGPR> RRTX_DEV void push_if_not_less(float value, float thr, uint32_t &cnt, uint32_t lane_lds_addr, uint32_t *my_cand, int k0)
{
#if RRTX_ASM_PUSH
    uint32_t addr, tmp;
    uint64_t save;
    if (THR_IN_VGPR)
        RRTX_PUSH_ASM("v_cmp_ngt_f32", "v");
    else
        RRTX_PUSH_ASM("v_cmp_ngt_f32", "s");
#else
    if (!(value < thr)) {
        my_cand[cnt * 64] = (uint32_t)(k0 + U);
        cnt += 1;
    }
#endif
}
template <int U, bool THR_IN_VGPR> RRTX_DEV void push_if_not_less(double value, double thr, uint32_t &cnt, uint32_t lane_lds_addr, uint32_t *my_cand, int k0)
{
#if RRTX_ASM_PUSH
    uint32_t addr, tmp;
    uint64_t save;
    if (THR_IN_VGPR)
        RRTX_PUSH_ASM("v_cmp_ngt_f64", "v");
    else
        RRTX_PUSH_ASM("v_cmp_ngt_f64", "s");
#else
    if (!(value < thr)) {
        my_cand[cnt * 64] = (uint32_t)(k0 + U);
        cnt += 1;
    }
#endif
}

// LDSMODE 3 (the filter on the matrix cores): slots of a lane's list of (block of 16 spheres, 16 sign bits) entries
constexpr int kMfSlots = 8;
#ifndef RRTX_MF_BALANCE
#define RRTX_MF_BALANCE 1 // the pairs the filter lets through are shared out evenly over the wave's lanes before the exact test
#endif
#ifndef RRTX_MF_PRIO
#define RRTX_MF_PRIO 0 // > 0: s_setprio of a wave inside phase 1; < 0: of a wave outside it (experiments: 1, 3, -1 all within 0.1 % on fp32, 1 % on fp64)
#endif
constexpr int kMfQueue = 256; // ... through a queue of this many pairs (more than that: every lane tests what it listed)
// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// The launch parameters as they lie in the kernel-argument segment, behind a pointer the optimiser
// cannot see through: fields read via cold_params() are s_load-ed where they are used (camera, task
// decoding: once per sample) instead of being hoisted out of the render loop into SGPRs that the scan
// needs — the kernel is SGPR-bound, and every spilled SGPR is a v_readlane in somebody's way.
template <typename F> RRTX_DEV const RRTX_CONST_AS KernelParams<F> *cold_params()
{
    const RRTX_CONST_AS KernelParams<F> *p = (const RRTX_CONST_AS KernelParams<F> *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Work units of a parked item (written by the render kernel, P.tail_units = item << 3 | k): unit 0
// continues the item's partial sum over all but the last kTailSplit - 1 remaining samples of its task,
// units k >= 1 are those last samples one by one; the result of unit k goes to tail_rad[item][k] and
// tail_sum_kernel adds them in sample order.
template <typename F>
RRTX_DEV void load_unit(const KernelParams<F> &P, uint32_t u, uint32_t &task, int &s_cur, int &s_end, bool &need_ray, bool &single, uint32_t &out_index, V3<F> &acc, Path<F> &path, Rng &rng)
{
    const uint32_t item = u >> 3, k = u & 7u;
    const TailItem<F> it = P.tail_items[item];
    task = it.task;
    int px_i, px_j, s_first, s_task_end;
    task_decode<F>(P, task, px_i, px_j, s_first, s_task_end);
    const int remaining = s_task_end - it.s_cur;
    const int n0 = remaining > kTailSplit - 1 ? remaining - (kTailSplit - 1) : 1;
    if (k == 0) {
        s_cur = it.s_cur, s_end = it.s_cur + n0;
        need_ray = it.need_ray != 0;
        acc = mk<F>(it.acc[0], it.acc[1], it.acc[2]);
        path.o = mk<F>(it.o[0], it.o[1], it.o[2]);
        path.d = mk<F>(it.d[0], it.d[1], it.d[2]);
        path.tm = it.tm;
        path.atten = mk<F>(it.atten[0], it.atten[1], it.atten[2]);
        path.depth = it.depth;
        rng.k0 = it.k0, rng.k1 = it.k1, rng.n = it.n;
    }
    else {
        s_cur = it.s_cur + n0 + (int)k - 1, s_end = s_cur + 1;
        need_ray = true;
    }
    single = k != 0;
    out_index = item * (uint32_t)kTailSplit + k;
}

// A sample's radiance is known: rrt.cu:115, `pixel_color += ray_color(...)`, then the next sample or the end of the task.
// per_sample launches (the reference's own summation order at the speed of small work items): nothing is added here - the sample's
// radiance goes to its own slot, [pixel][sample][3], and finalize_kernel forms the pixel's running sum in sample order.
template <typename F, bool RESUME, bool PLAIN = false> RRTX_DEV void finish_sample(const KernelParams<F> &P, const V3<F> &radiance, uint32_t task, int &s_cur, int s_end, bool single, uint32_t out_index, V3<F> &acc, bool &need_task, bool &need_ray)
{
    const auto &C = *cold_params<F>();
    if (!PLAIN && C.per_sample) {
        F *o = C.out + ((size_t)task_pixel<F>(C, task) * (size_t)C.spp + (size_t)s_cur) * 3;
        o[0] = radiance.x, o[1] = radiance.y, o[2] = radiance.z;
        s_cur += 1;
        if (s_cur == s_end)
            need_task = true;
        else
            need_ray = true;
        return;
    }
    acc = (RESUME && single) ? radiance : vadd<F>(acc, radiance);
    s_cur += 1;
    if (s_cur == s_end) {
        F *o = RESUME ? P.tail_rad + (size_t)out_index * 3 : task_slot<F>(C, task);
        o[0] = acc.x;
        o[1] = acc.y;
        o[2] = acc.z;
        need_task = true;
    }
    else
        need_ray = true;
}

// Records the scan reads: the fp32 filter table, or (FILTER = false) the exact-test table in F.
template <typename F, bool FILTER> struct ScanType {
    typedef F type;
    static RRTX_DEV const SphereHot<F> *table(const KernelParams<F> &P) { return P.sph_hot; }
};
template <typename F> struct ScanType<F, true> {
    typedef float type;
    static RRTX_DEV const SphereHot<float> *table(const KernelParams<F> &P) { return P.sph_filter; }
};

// LDSMODE: where the scan reads its sphere records from.  0 = scalar loads only; 1 = blocks alternate
// between scalar loads and broadcast reads of a copy in LDS; 2 = LDS only.  At 8 VALU per test the
// scalar data cache (shared by CUs, ~4.5 B/clk) is the binding unit, which is what the LDS copy relieves.
#ifndef RRTX_LIST_WAVES_F64
#define RRTX_LIST_WAVES_F64 5 // waves per SIMD the fp64 list-scan variants are compiled for (94 VGPRs, 12 spilled: 92.4 vs 102.0 ms; 6: 100.1)
#endif
#ifndef RRTX_MF_WAVES_F64
#define RRTX_MF_WAVES_F64 (RRTX_MF_BLOCK_THREADS_F64 == 768 ? 3 : 2) // waves per SIMD the fp64 matrix-core scan is compiled for: what its LDS lets be resident
#endif
#ifndef RRTX_ACCEL_WAVES_F64
#define RRTX_ACCEL_WAVES_F64 4 // ... and the fp64 ones (127 VGPRs: 66.9 vs 72.5 ms at spp 504; the fp64 list scan takes 128.2)
#endif
#ifndef RRTX_ACCEL_POLL_MASK
#define RRTX_ACCEL_POLL_MASK 3u
#endif
#ifndef RRTX_WALK_SLICE
#define RRTX_WALK_SLICE 0 // > 0 overrides GridRec::walk_slice, the cells a lane walks per iteration of the render loop (final.txt: 2 / 3 / 4 / 6 / all: 49.4 / 45.5 / 45.1 / 46.2 / 50.0 ms)
#endif
#ifndef RRTX_DENSE_WAVES
#define RRTX_DENSE_WAVES 4 // ... the variants that pair (ray, entry) densely (scenes with triangles / moving spheres): their launches are bound by the latency of dependent loads, not by occupancy, and at 80 registers they spilled 60
#endif
#ifndef RRTX_DENSE_WAVES_F64
#define RRTX_DENSE_WAVES_F64 3 // (27 072-triangle mesh 600x400 spp 16, fp64: 2 / 3 / 4 waves per SIMD 7.7 / 6.7 / 10.5 ms; fp32: 3 / 4 / 5 -> 5.5 / 5.85 / 9.0)
#endif
#ifndef RRTX_ACCEL_WAVES
#define RRTX_ACCEL_WAVES 6 // waves per SIMD the accelerated fp32 variants are compiled for (80 VGPRs: 43.6 vs 45.5 ms without the limit)
#endif
// ACCEL: 0 = every segment is scanned; 1 / 2 = accelerated closest hit (accel_closest_hit) with the grid
// and the exact-test records read from HBM / from a copy in LDS, the scan being the fallback for the
// rays the grid is not proven for.
// RESUME: a second pass of the accelerated variants over the work units the first pass parked when the
// queue ran out (lanes take up paths where they were left, densely packed again); results go to
// tail_rad, nothing is parked again: an accelerated iteration is cheap enough to run paths to their end.
// The ray of lane `src` against every primitive, by all 64 lanes of the wave: lane l tests primitives l, l + 64, ...
// with the exact test; a butterfly picks the winner by consider()'s order-independent rule, which is the answer
// of the sequential scan for any finite ray.  Every lane returns the result.
template <typename F, bool SO = false, typename HotTab> __device__ __forceinline__ HitInfo<F> wave_exact_scan(const KernelParams<F> &P, HotTab hot, const Path<F> &path, int src, int lane, F t_min)
{
    const int msph_base = P.n_sph_padded, tri_base = P.n_sph_padded + P.n_msph; // (the loop's bounds: the real ones)
    Path<F> rp;
    rp.o = mk<F>(__shfl(path.o.x, src), __shfl(path.o.y, src), __shfl(path.o.z, src));
    rp.d = mk<F>(__shfl(path.d.x, src), __shfl(path.d.y, src), __shfl(path.d.z, src));
    rp.tm = __shfl(path.tm, src);
    rp.atten = mk<F>(0, 0, 0), rp.depth = 0;
    const F ra = vlen2<F>(rp.d);
    HitInfo<F> hb = {Limits<F>::inf(), -1};
    PendingRoot<F> pend = {-1, 0, 0};
    for (int idx = lane; idx < tri_base + P.n_tri; idx += 64) {
        if (idx >= P.n_sph && idx < msph_base) continue; // (padding records)
        test_primitive<F, SO>(P, hot, idx, rp, ra, t_min, hb, pend);
    }
    resolve_pending<F>(pend, ra, t_min, SO ? kNoTriangles : tri_base, hb);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const F ot = __shfl_xor(hb.t, off);
        const int oi = __shfl_xor(hb.idx, off);
        if (oi >= 0) consider<F>(ot, oi, SO ? kNoTriangles : tri_base, hb);
    }
    return hb;
}

// The reference's sequential scan (hittable_list.h:95-117) by one lane: every primitive in list order with the exact test
// and the running closest_so_far.  What the accelerated variants fall back to for the rays the unordered rule is not
// proven for (non-finite or absurd components: practically never), and what the VERIFY builds compare against.
template <typename F> __device__ __forceinline__ HitInfo<F> sequential_closest_hit(const KernelParams<F> &P, const Path<F> &path, F a, F t_min)
{
    HitInfo<F> full = {Limits<F>::inf(), -1};
    const int msph_base = P.n_sph_padded, tri_base = P.n_sph_padded + P.n_msph;
    for (int q = 0; q < P.n_sph; ++q) {
        const SphereHot<F> gq = P.sph_hot[q];
        refine_sphere<F, true>(gq.cx, gq.cy, gq.cz, gq.r2, path, a, t_min, q, full);
    }
    for (int q = 0; q < P.n_msph; ++q) {
        const MovingSphereRec<F> ms = P.msph[q];
        const V3<F> cen = msphere_center<F>(ms, path.tm);
        refine_sphere<F, true>(cen.x, cen.y, cen.z, ms.r2, path, a, t_min, msph_base + q, full);
    }
    for (int q = 0; q < P.n_tri; ++q) {
        F tt;
        if (triangle_test<F, true>(P.tri[q], path, t_min, full.t, tt)) {
            full.t = tt;
            full.idx = tri_base + q;
        }
    }
    return full;
}

// ---------------------------------------------------------------------------------------------
// Dense (ray, entry) pairs (accelerated variants).  In the grid walk every lane used to run its own loops - over the cells of
// its slice, over each cell's entries - and the wave paid for the LONGEST of them: 2.8 trips of the test loop per cell for 1.9
// entries per lane and cell, with 20 of 64 lanes left in the average trip (35 % lane utilisation over the whole kernel, round 2).
// Here every lane only LISTS what it wants tested this iteration - the entries of the next cells of its walk
// (accel_walk_prepare: up to four (first, count) ranges of cell_prims), or, for a fresh camera ray, its pixel's candidate list
// (+ every moving sphere and triangle, as the LIST passes of the list-scan kernel test them) - and the wave then shares all
// those (ray, entry) pairs out evenly: pair p = base + lane of a trip belongs to the lane whose range [off, off + n) of the
// prefix sum holds p (found with one mark per owner in LDS and a prefix maximum), its ray comes across with ds_bpermute, and
// the lane evaluates the reference's exact test for it (sphere.h:33-49, moving_sphere.h:27-49, triangle.h:35-71) with the owner's
// operands in the owner's arithmetic - the same operations in the same order as test_primitive(), so the same bits.  A pair that
// yields a hit (t >= t_min; the window's far end is the owner's business) is folded into the owner's slot in LDS with an atomic
// minimum under consider()'s order - the smallest t; at equal t the sphere-like primitive of the highest index, a triangle only
// if no sphere has that t, and then the one of the lowest index (rrtx_path.h) - which is the answer of the sequential scan whatever
// order the hits arrive in.  fp32: ONE ds_min_u64 on (bits of t << 32 | ~rank) - t is positive, so its bits order like its value.
// fp64: a ds_min_u64 on the bits of t, then the lanes whose t IS the owner's minimum raise the owner's rank with a ds_max_u32
// (the rank slot cleared first by whoever lowered the minimum in this trip).  The owner reads its slot when the trips are over.
// All 64 lanes call this, converged.  marks, ranks: 64 dwords of LDS each, keys: 64 qwords, this wave's.
// ---------------------------------------------------------------------------------------------
template <typename F, bool SO, int CAP, typename HotTab, typename PrimTab>
__device__ __forceinline__ void dense_candidates(const KernelParams<F> &P, const HotTab &hot, const PrimTab &cell_prims, const uint16_t *plist, const Path<F> &path, F a, const WalkRanges &R,
                                                  F t_min, bool is_list, uint32_t n_own, int lane, uint32_t *marks, uint32_t *ranks, unsigned long long *keys)
{
    // "All 64 lanes, converged" is enforced, not assumed: the prefix sums (DPP), the read of lane 63 and the bpermutes below take operands from
    // every lane, and a lane masked off would feed them whatever its registers happen to hold.  A wave that arrives here short of lanes counts
    // a fault (rrtx_stats.convergence_faults; every GPU test asserts 0), and the trip count is bounded by what 64 lanes can list whatever the sum says.
    if (__builtin_expect(__builtin_amdgcn_read_exec() != ~0ull, 0)) atomicAdd(&P.counters[6], 1ull);
    const uint32_t incl = wave_scan_add(n_own), off = incl - n_own;
    constexpr uint32_t kMostPairs = 64u * kDenseCellMax * (uint32_t)kDenseRanges;
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    total = total < kMostPairs ? total : kMostPairs;
    keys[lane] = ~0ull;
    if (sizeof(F) == 8) ranks[lane] = 0u;
    const int msph_base = P.n_sph_padded, tri_base = P.n_sph_padded + P.n_msph;
    const uint32_t off_and_kind = off | (is_list ? 0x80000000u : 0u);
    for (uint32_t base = 0; base < total; base += 64u) {
        // whose pair is base + lane?  Every owner marks the first slot of its range inside this window; a prefix maximum
        // carries the mark to the slots behind it (ranges are disjoint and in lane order)
        marks[lane] = 0u;
        if (n_own != 0u && off < base + 64u && off + n_own > base) marks[off > base ? off - base : 0u] = (uint32_t)lane + 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (LDS operations of a wave execute in order: this only keeps the compiler from moving them)
        const uint32_t m = wave_scan_max(marks[lane]);
        const uint32_t p = base + (uint32_t)lane;
        const bool live = p < total;
        const int owner = live ? (int)m - 1 : lane;
        const uint32_t o_ok = (uint32_t)__shfl((int)off_and_kind, owner), o_cnt = (uint32_t)__shfl((int)R.cnt, owner);
        const uint32_t o_b0 = (uint32_t)__shfl((int)R.beg0, owner), o_b1 = (uint32_t)__shfl((int)R.beg1, owner), o_b2 = (uint32_t)__shfl((int)R.beg2, owner), o_b3 = (uint32_t)__shfl((int)R.beg3, owner);
        Path<F> rp;
        rp.o = mk<F>(__shfl(path.o.x, owner), __shfl(path.o.y, owner), __shfl(path.o.z, owner));
        rp.d = mk<F>(__shfl(path.d.x, owner), __shfl(path.d.y, owner), __shfl(path.d.z, owner));
        rp.tm = SO ? (F)0 : __shfl(path.tm, owner);
        rp.atten = mk<F>(0, 0, 0), rp.depth = 0;
        const F ra = __shfl(a, owner);
        bool keep = false;
        F t_hit = 0;
        uint32_t rank_u = 0; // consider()'s rank, biased to unsigned: spheres by index, triangles below all spheres and by descending index
        if (live) {
            const uint32_t j = p - (o_ok & 0x7FFFFFFFu);
            const uint32_t e = walk_range_entry(o_b0, o_b1, o_b2, o_b3, o_cnt, j);
            int idx;
            if (!(o_ok & 0x80000000u))
                idx = (int)cell_prims[e];
            else if (j < (o_cnt & 255u))
                idx = (int)plist[(size_t)o_b0 * kPlistStride + 1u + j]; // range 0 of a camera ray: its pixel's list (beg0 = the pixel)
            else
                idx = (int)e; // ranges 1, 2 of a camera ray: every moving sphere, every triangle
            rank_u = (uint32_t)((SO || idx < tri_base) ? idx : -idx - 1) ^ 0x80000000u;
            if (SO || idx < tri_base) {
                F cx, cy, cz, r2;
                if (SO || idx < msph_base) {
                    const SphereHot<F> g = hot[idx];
                    cx = g.cx, cy = g.cy, cz = g.cz, r2 = g.r2;
                }
                else {
                    const MovingSphereRec<F> ms = P.msph[idx - msph_base];
                    const V3<F> cen = msphere_center<F>(ms, rp.tm);
                    cx = cen.x, cy = cen.y, cz = cen.z, r2 = ms.r2;
                }
                const F ocx = rp.o.x - cx, ocy = rp.o.y - cy, ocz = rp.o.z - cz; // sphere.h:35-40
                const F half_b = ocx * rp.d.x + ocy * rp.d.y + ocz * rp.d.z;
                const F c = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;
                const F disc = half_b * half_b - ra * c;
                keep = !(disc < 0);
                if (keep) { // sphere.h:41-49 without the dependence on the scan order (resolve_pending)
                    const F sq = fsqrt(disc);
                    t_hit = (-half_b - sq) / ra;
                    if (t_hit < t_min) {
                        t_hit = (-half_b + sq) / ra;
                        keep = !(t_hit < t_min);
                    }
                }
            }
            else
                keep = triangle_test<F, true>(P.tri[idx - tri_base], rp, t_min, Limits<F>::inf(), t_hit);
        }
        if (sizeof(F) == 4) {
            if (keep) atomicMin(&keys[owner], ((unsigned long long)__float_as_uint((float)t_hit) << 32) | (unsigned long long)(~rank_u)); // ds_min_u64
        }
        else {
            const unsigned long long tb = (unsigned long long)__double_as_longlong((double)t_hit);
            const unsigned long long before = keys[owner];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (keep) atomicMin(&keys[owner], tb);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool wins = keep && keys[owner] == tb; // this pair's t is the owner's minimum so far
            if (wins && tb < before) ranks[owner] = 0u;  // ... a new one: what the slot says belongs to a larger t
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (wins) atomicMax(&ranks[owner], rank_u + 1u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
constexpr uint32_t kCoopWait = 0xFFFFFFFFu, kCoopDone = 0xFFFFFFFEu; // walk_cell of a far ray before / after the wave's scan (cells use 30 bits)
template <typename F, bool FILTER, int LDSMODE, bool VERIFY, int ACCEL, bool RESUME = false, int SOV = 0> __global__ void __launch_bounds__((LDSMODE == 3 ? mf_block_threads(sizeof(F)) : kBlockThreads), (ACCEL != 0 ? (sizeof(F) == 4 ? ((SOV & 3) != 0 ? RRTX_ACCEL_WAVES : RRTX_DENSE_WAVES) : ((SOV & 3) != 0 ? RRTX_ACCEL_WAVES_F64 : RRTX_DENSE_WAVES_F64)) : (LDSMODE == 3 ? (sizeof(F) == 4 ? 4 : RRTX_MF_WAVES_F64) : (sizeof(F) == 8 ? RRTX_LIST_WAVES_F64 : 1)))) render_kernel(const KernelParams<F> P)
{
    // SOV: 0 = scenes of every kind; 1 = spheres alone (SO: the branches that tell the kinds apart are compiled out); 2 = spheres alone AND every fresh sample of the
    // launch is a first-bounce record (KernelParams::first is set: first_bounce_kernel ran) - camera rays and LIST passes are compiled out of the loop as well
    // (final.txt use_bvh 36.3 -> 35.6 ms, fp64 56.25 -> 55.6 against the same loop deciding at run time)
    // (the matrix-core list scan is chosen for scenes of spheres alone only: rrtx_api.cpp.  C3 49.22 -> 48.66 ms, C4 65.3 -> 64.9, C2 1.125 -> 1.106)
    // SOV & 4 (kPlain): the launch has no single-sample tasks at its end, a queue order (the sky split) and per-task sums - decided once per launch, not asked at
    // every task and sample (C3 48.63 -> 48.04 ms, C4 64.8 -> 63.8, use_bvh 35.69 -> 35.59 / 55.2 -> 54.6).  Never in a resume pass.
    constexpr bool SO = (SOV & 3) != 0 || LDSMODE == 3, kFirstAlways = (SOV & 3) == 2, kPlain = (SOV & 4) != 0 && !RESUME;
    constexpr int kBT = LDSMODE == 3 ? mf_block_threads(sizeof(F)) : kBlockThreads, kWPB = kBT / 64; // threads, waves of a block of this variant
    // candidate slots per lane for the scan: the accelerated variants scan one segment in a hundred
    // thousand and rather keep the LDS for a sixth block per CU
    constexpr int kCap = ACCEL != 0 ? 8 : (LDSMODE == 3 && sizeof(F) == 8 ? 24 : kCandCap); // (LDSMODE 3: 2 KB of pair lists + 64 ray records)
    __shared__ __attribute__((aligned(16))) uint32_t cand_lds[kWPB][kCap][64];
    // accelerated variants: the dense (ray, entry) pairing's owner marks and candidate counters (dense_candidates)
    // Measured (round 3): the dense pairing wins where a test is dear and a lane's loops are long - scenes with triangles or
    // moving spheres: 27 072 triangles 600x400 spp 16: 20.7 -> 5.85 ms (fp64 21.5 -> 6.7), frames identical - and loses where the
    // walk was cheap to begin with: final.txt (488 spheres, tables in LDS) 37.6 -> 46.5 ms, 40 000 spheres 8.5 -> 9.4 ms - there a
    // lane's listing of its cells costs more than the per-lane walk it replaces (EXPERIMENTS.md).  So: scenes of spheres alone keep
    // the per-lane walk, everything else is paired densely.
#ifndef RRTX_DENSE_ALL
#define RRTX_DENSE_ALL 0 // 1: scenes of spheres alone are paired densely too (experiments)
#endif
    constexpr bool kDensePairs = ACCEL != 0 && (!SO || RRTX_DENSE_ALL);
    constexpr int kDenseLds = kDensePairs ? 64 : 1;
    __shared__ uint32_t dense_marks[kWPB][kDenseLds], dense_ranks[kWPB][kDenseLds];
    __shared__ unsigned long long dense_keys[kWPB][kDenseLds];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[]; // LDSMODE != 0: n_sph_padded scan records; ACCEL == 2: grid
    typedef typename ScanType<F, FILTER>::type ST; // precision of the scan's records: the filter is fp32 for every F
    SphereHot<ST> *const sph_lds = (SphereHot<ST> *)dyn_lds;
    if (LDSMODE == 1 || LDSMODE == 2) {
        const SphereHot<ST> *src = ScanType<F, FILTER>::table(P);
        for (int i = threadIdx.x; i < P.n_sph_padded; i += kBT) sph_lds[i] = src[i];
        __syncthreads();
    }
    // LDSMODE == 3: the filter on the matrix cores - the spheres' f16 operands (64 bytes each, rrtx_pack.h: pack_mf_table) in LDS, the candidate
    // lists as 32 x 16 bits per lane (other lanes push onto them: the counters live in LDS too)
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    U4 *const mf_lds = (U4 *)dyn_lds;
    __shared__ unsigned long long mf_keys[kWPB][LDSMODE == 3 ? 64 : 1];
    __shared__ uint32_t mf_ranks[kWPB][LDSMODE == 3 && sizeof(F) == 8 ? 64 : 1];
    __shared__ uint16_t mf_queue[kWPB][LDSMODE == 3 && RRTX_MF_BALANCE ? kMfQueue : 1]; // (ray, sphere) pairs on their way to the exact test
    if (LDSMODE == 3) {
        const U4 *src = (const U4 *)P.mf_table;
        for (int i = threadIdx.x; i < ((P.n_sph_padded + 31) & ~31) * 4; i += kBT) mf_lds[i] = src[i]; // (whole blocks of 32 spheres)
        __syncthreads();
    }
    static_assert(ACCEL == 0 || LDSMODE == 0, "the accelerated variants scan from scalar loads");
    static_assert(LDSMODE != 3 || (FILTER && kCap * 256 >= 2048 + 64 * 8 * (int)sizeof(F) && kCap >= 16), "the matrix-core filter's staging, pair lists and ray records live in the wave's candidate slots");
    // ACCEL == 2: [exact-test records][cell_start][cell_prims] in LDS
    SphereHot<F> *const hot_lds = (SphereHot<F> *)dyn_lds;
    uint32_t *const cell_start_lds = (uint32_t *)(hot_lds + P.n_sph_padded);
    GridPrim *const cell_prims_lds = (GridPrim *)(cell_start_lds + P.n_grid_cells + 1);
    if (ACCEL == 2) {
        for (int i = threadIdx.x; i < P.n_sph_padded; i += kBT) hot_lds[i] = P.sph_hot[i];
        for (int i = threadIdx.x; i <= P.n_grid_cells; i += kBT) cell_start_lds[i] = P.grid_cell_start[i];
        for (int i = threadIdx.x; i < P.n_grid_prims; i += kBT) cell_prims_lds[i] = P.grid_cell_prims[i];
        __syncthreads();
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#if RRTX_MF_PRIO < 0
    if (LDSMODE == 3) __builtin_amdgcn_s_setprio(-(RRTX_MF_PRIO));
#endif
    uint32_t *const my_cand = &cand_lds[wave][0][lane]; // slot s at my_cand[s * 64]: bank == lane, conflict-free
    const uint32_t my_cand_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)my_cand; // LDS byte address
    const ST zero_sgpr = (ST)0;

    typedef const RRTX_CONST_AS SphereHot<ST> *HotPtr; // constant address space => s_load for uniform indices
    const HotPtr sph_scalar = (HotPtr)ScanType<F, FILTER>::table(P);

    const F t_min = (F)0.001; // rrt.cpp:32 typing
    const int n_sph = P.n_sph, n_sph_pad = P.n_sph_padded, n_msph = SO ? 0 : P.n_msph, n_tri = SO ? 0 : P.n_tri; // (SO: their loops fold away)
    const int msph_base = n_sph_pad, tri_base = n_sph_pad + n_msph;

    // wave-uniform task pool
    uint32_t pool_next = 0, pool_end = 0;
    bool queue_dry = false;  // this wave's own pull came back empty
    bool queue_over = false; // the global cursor has been seen past the end
    uint32_t cursor_seen = 0, loop_count = 0;
    const uint32_t n_waves = gridDim.x * (uint32_t)kWPB;
    int dry_iters = 0; // loop iterations since this wave saw the queue over

    // lane state
    bool alive = true, need_task = true, need_ray = false;
    uint32_t task = 0;
    int s_cur = 0, s_end = 0; // (the pixel is re-derived from `task` at each camera ray: two VGPRs = one wave of occupancy)
    V3<F> acc = mk<F>(0, 0, 0);
    Path<F> path;
    path.o = path.d = path.atten = mk<F>(0, 0, 0);
    path.tm = 0;
    path.depth = 0;
    Rng rng = {0, 0, 0};
    uint32_t out_index = 0; // RESUME: where the unit's result goes in tail_rad
    bool single = false;    // RESUME: the result is the sample itself (units k >= 1), not a running sum
    const uint32_t n_units_in = RESUME ? P.tail_count[2] : 0u;
    // RESUME: the FIRST 64 units of every wave are dealt, not pulled.  The waves of a pass all start within microseconds of one another, atomics on one word are served
    // one after the other (88 per microsecond: MI355X_MICROARCH.md), and what a launch parks rarely fills a tenth of the grid: the few waves with work queued behind
    // thousands that only came to find the cursor past the end.  Dealt, no wave of an ordinary pass touches the cursor at all (it counts from `dealt` on): C3 shard of 8
    // 7.25 -> 7.01 ms, C2 1.15 -> 1.13, fp64 mesh 6.72 -> 6.52.  (The same for the render kernel's first batch costs a full frame 1.5 % and buys C2 another 2 %: not
    // done.  EXPERIMENTS.md, round 4.)
    uint32_t dealt = 0;
    if (RESUME) {
        const uint32_t wave_id = blockIdx.x * (uint32_t)kWPB + (uint32_t)wave;
        dealt = n_waves * 64u;
        pool_next = wave_id * 64u < n_units_in ? wave_id * 64u : n_units_in;
        pool_end = n_units_in - pool_next < 64u ? n_units_in : pool_next + 64u;
        if (dealt >= n_units_in) queue_dry = queue_over = true; // (everything was dealt)
    }
    uint32_t n_segments = 0, n_candidates = 0, n_scanned = 0;
    uint32_t n_walk_cells = 0, n_walk_pairs = 0; // dense variants: cells stepped through, (ray, entry) pairs tested
#ifdef RRTX_RESUME_DIAG
    const unsigned long long resume_t0 = __builtin_amdgcn_s_memtime();
#endif
    uint32_t plist_count = 0xFFFFu; // header of the current pixel's camera-ray list
    // ACCEL: a grid walk in progress (see accel_closest_hit)
    bool in_walk = false;
    uint32_t walk_cell = 0;
    F walk_t_out = 0;
    HitInfo<F> best = {Limits<F>::inf(), -1}; // (lives across iterations only while a walk is in progress)
    int list_passes_done = 0;        // wave-uniform: consecutive LIST passes so far
#ifdef RRTX_DIAG // timing diagnostics (never in the product build): per-wave real-time stamps
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long diag_dry = 0;
    uint32_t diag_iters = 0, diag_iters_dry = 0, diag_pulls = 0, diag_pull_iter = 0, diag_pull_base = 0;
    unsigned long long diag_pull_t = 0;
#endif

#ifdef RRTX_SECTION_DIAG // developer builds: clock cycles of a wave per section of the loop -> counters[16 + k]
    unsigned long long dense_dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long sec_cycles[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sec_t = __builtin_amdgcn_s_memtime();
    int sec_cur = 0;
#define RRTX_SEC(k)                                                                                                                        \
    do {                                                                                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                                  \
        sec_cycles[sec_cur] += now_ - sec_t;                                                                                           \
        sec_t = now_, sec_cur = (k);                                                                                                   \
    } while (0)
#else
#define RRTX_SEC(k) ((void)0)
#endif
    for (;;) {
        RRTX_SEC(0); // hand-out, polling, hand-off
#ifdef RRTX_DIAG
        diag_iters += 1;
        if (queue_over) {
            if (!diag_dry) diag_dry = __builtin_amdgcn_s_memrealtime();
            diag_iters_dry += 1;
        }
#endif
        // ---------------- the global cursor: is the queue over, and how large a batch to pull ------
        // The first wave whose pull comes back empty raises a flag; every wave polls it, also those
        // that need no task: a wave that is slow (the youngest waves
        // of a SIMD get the fewest issue slots) or holds long tasks would otherwise work through its
        // pool long after everyone else has left.
        if (!RESUME && !queue_over && P.handoff_lanes > 0 && (loop_count & (ACCEL != 0 ? RRTX_ACCEL_POLL_MASK : 3u)) == 0u) { // (every 4th iteration: free; every iteration: 7 %; nobody hands off without a tail kernel)
            // (a flag on a line of its own: reading the cursor itself, which every pull hits with an
            // atomic, costs ~30 us a poll)
            uint32_t over = 0;
            if (lane == 0) over = __hip_atomic_load(P.queue + kQueueOverFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            queue_over = __builtin_amdgcn_readfirstlane(over) != 0u;
        }
        loop_count += 1;
        if (ACCEL != 0) {
            // ---------------- far rays: an exact scan by the whole wave ---------------------------------
            // A ray from beyond the grid's range that touches its fattened box has to be tested against every
            // primitive.  One lane doing that alone holds up its wave for N tests (0.3 ms for 40 000 spheres,
            // and a path that wanders about out there needs one per bounce: the resume pass of such a scene took
            // longer than the render).  Here, where all 64 lanes are active, the wave takes such rays one at a time:
            // lane l tests primitives l, l + 64, ... with the exact test, and a butterfly picks the winner
            // by consider()'s order-independent rule - the answer of the sequential scan.
            uint64_t waiting = __ballot(in_walk && walk_cell == kCoopWait);
            while (__builtin_expect(waiting != 0ull, 0)) { // (rare: marked so, for the register allocator to spill in here and not in the loop)
                const int src = (int)__builtin_ctzll(waiting);
                waiting &= waiting - 1ull;
                HitInfo<F> hb;
                if (ACCEL == 2)
                    hb = wave_exact_scan<F, SO>(P, hot_lds, path, src, lane, t_min);
                else
                    hb = wave_exact_scan<F, SO>(P, P.sph_hot, path, src, lane, t_min);
                if (lane == src) best = hb, walk_cell = kCoopDone, n_scanned += 1;
            }
        }
        // ---------------- task hand-out: wave64 ballot + prefix popcount -------------------------
        uint64_t want = __ballot(need_task);
        while (want != 0ull) {
            if (RESUME && pool_next == pool_end) {
                if (queue_dry) break;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.tail_count + 1, 64u);
                base = __builtin_amdgcn_readfirstlane(base) + dealt;
                if (base >= n_units_in) {
                    queue_dry = queue_over = true;
                    break;
                }
                pool_next = base;
                pool_end = n_units_in - base < 64u ? n_units_in : base + 64u;
            }
            if (!RESUME && pool_next == pool_end) {
                if (queue_dry) break;
                // guided batches: half of an even share of what is left of the region (chunk tasks, then
                // single-sample tasks), so that the pools waves are left with shrink towards its end
                uint32_t batch;
                const uint32_t chunk_region_end = P.taper_task_base < P.total_tasks ? P.taper_task_base : P.total_tasks; // (the sky split: the queue ends before the last chunk task)
                if (cursor_seen < chunk_region_end) {
                    batch = (chunk_region_end - cursor_seen) / (2u * n_waves);
                    batch = batch < kTaskBatchMin ? kTaskBatchMin : (batch > kTaskBatch ? kTaskBatch : batch);
                }
                else {
                    batch = (P.total_tasks > cursor_seen ? P.total_tasks - cursor_seen : 0u) / (2u * n_waves);
                    batch = batch < kTaskBatch ? kTaskBatch : (batch > kTaperBatch ? kTaperBatch : batch);
                }
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.queue, batch);
                base = __builtin_amdgcn_readfirstlane(base);
                cursor_seen = base + batch;
                if (base >= P.total_tasks) {
                    if (lane == 0 && !queue_over) __hip_atomic_store(P.queue + kQueueOverFlag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    queue_dry = queue_over = true;
                    break;
                }
                pool_next = base;
                pool_end = (P.total_tasks - base < batch) ? P.total_tasks : base + batch;
#ifdef RRTX_DIAG
                diag_pulls += 1, diag_pull_iter = diag_iters, diag_pull_base = base, diag_pull_t = __builtin_amdgcn_s_memrealtime();
#endif
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u)); // set bits below this lane
            if (need_task && rank < avail) {
                need_task = false;
                if (RESUME) {
                    load_unit<F>(P, P.tail_units[pool_next + rank], task, s_cur, s_end, need_ray, single, out_index, acc, path, rng);
                    plist_count = 0xFFFFu; // a path in flight has no camera-ray list (a fresh camera ray fetches its own)
                }
                else {
                    task = queue_task<kPlain>(*cold_params<F>(), pool_next + rank);
                    int pi, pj;
                    task_decode<F, kPlain>(*cold_params<F>(), task, pi, pj, s_cur, s_end);
                    acc = mk<F>(0, 0, 0);
                    need_ray = true;
                }
            }
            const uint32_t wanted = (uint32_t)__popcll(want);
            pool_next += wanted < avail ? wanted : avail;
            want = __ballot(need_task);
        }
        if (need_task) { // queue exhausted: this lane retires
            alive = false;
            need_task = false;
        }
        {
            const uint64_t live = __ballot(alive);
            if (live == 0ull) break;
            // ---------------- hand-off to the tail kernel --------------------------------------------
            // Once the queue is over, a wave runs on with fewer and fewer live lanes (in the end the
            // 0.36 % of paths that bounce 50 times), and a lone wave needs ~10 us per segment of the
            // lane-per-ray scan: that tail cost ~6 ms per launch whatever the frame size.  Below
            // `handoff_lanes` live lanes the wave parks its unfinished work items (pixel, sample, ray,
            // attenuation, RNG position, partial sum) in HBM and exits; tail_kernel finishes them (the
            // accelerated variants: a RESUME pass of this kernel).
            // ... or after `handoff_iters` more iterations, whatever is still alive — and whatever tasks
            // are left in its pool: the launch then ends a bounded time after the queue does.
            if (queue_over) dry_iters += 1;
            if (!RESUME && queue_over && P.handoff_lanes > 0 && pool_end - pool_next <= 64u && ((int)__popcll(live) <= P.handoff_lanes || dry_iters > P.handoff_iters)) {
                // (every lane still executes here; lane 0 speaks for the wave)
                auto park = [&](bool active, const TailItem<F> &it, int units) {
                    const uint64_t who = __ballot(active);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.tail_count, (uint32_t)__popcll(who));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(who >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)who, 0u));
                    if (active) P.tail_items[slot] = it;
                    // ... and the item's work units (see tail_kernel), first units first: those are the
                    // paths already in flight
                    if (!active) units = 0;
                    uint64_t has[kTailSplit];
                    uint32_t total = 0;
#pragma unroll
                    for (int k = 0; k < kTailSplit; ++k) {
                        has[k] = __ballot(units > k);
                        total += (uint32_t)__popcll(has[k]);
                    }
                    uint32_t ubase = 0;
                    if (lane == 0) ubase = atomicAdd(P.tail_count + 2, total);
                    ubase = __builtin_amdgcn_readfirstlane(ubase);
#pragma unroll
                    for (int k = 0; k < kTailSplit; ++k) {
                        if (units > k) P.tail_units[ubase + __builtin_amdgcn_mbcnt_hi((uint32_t)(has[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has[k], 0u))] = (slot << 3) | (uint32_t)k;
                        ubase += (uint32_t)__popcll(has[k]);
                    }
                };
                if (ACCEL != 0 && alive && in_walk) n_segments -= 1; // parked in mid-walk: the resume pass walks (and counts) this segment from its start
                TailItem<F> it;
                it.task = task, it.s_cur = s_cur, it.depth = path.depth, it.need_ray = need_ray ? 1u : 0u;
                it.k0 = rng.k0, it.k1 = rng.k1, it.n = rng.n, it.pad = 0;
                it.acc[0] = acc.x, it.acc[1] = acc.y, it.acc[2] = acc.z;
                it.o[0] = path.o.x, it.o[1] = path.o.y, it.o[2] = path.o.z;
                it.d[0] = path.d.x, it.d[1] = path.d.y, it.d[2] = path.d.z;
                it.tm = path.tm;
                it.atten[0] = path.atten.x, it.atten[1] = path.atten.y, it.atten[2] = path.atten.z;
                park(alive, it, s_end - s_cur < kTailSplit ? s_end - s_cur : kTailSplit);
                if (pool_next != pool_end) { // tasks nobody started
                    const bool mine = pool_next + (uint32_t)lane < pool_end;
                    TailItem<F> fresh = {};
                    int pi, pj, sf = 0, se = 0;
                    const uint32_t fresh_task = mine ? queue_task<kPlain>(*cold_params<F>(), pool_next + (uint32_t)lane) : 0u;
                    if (mine) task_decode<F, kPlain>(*cold_params<F>(), fresh_task, pi, pj, sf, se);
                    fresh.task = fresh_task, fresh.s_cur = sf, fresh.need_ray = 1u;
                    park(mine, fresh, se - sf < kTailSplit ? se - sf : kTailSplit);
                }
                break;
            }
        }

        RRTX_SEC(1); // camera rays
        if (alive && need_ray) {
            // ---------------- camera ray: rrt.cu:112-114, camera.h:31-38 --------------------------
            int pi, pj, sf, se;
            const auto &C = *cold_params<F>();
            task_decode<F, kPlain>(C, task, pi, pj, sf, se);
            // (the list-scan variants never meet a pre-pass - it was measured to gain them nothing - and are compiled without the record path: C3 49.66 -> 49.39 ms)
            const bool no_first = kFirstAlways ? false : (ACCEL == 0 ? true : C.first == nullptr);
            if (no_first) {
                need_ray = false;
                camera_ray<F>(C, pi, pj, s_cur, rng, path);
                // this pixel's camera-ray candidate list (header: count, or 0xFFFF = "scan everything")
                plist_count = C.plist ? (uint32_t)C.plist[(size_t)task_pixel<F, kPlain>(C, task) * kPlistStride] : 0xFFFFu;
            }
            else {
                // The first bounce of every sample was done before this kernel started (first_bounce_kernel: one lane per task, all lanes at work on neighbouring
                // samples of a pixel - camera ray, the pixel's candidate list, the primary hit's shading).  Here a fresh sample is a record: the path ended - its
                // radiance is added, on to the next sample -, or it goes on: the lane takes the scattered ray up at depth 1.  No camera ray is formed in this loop, no
                // LIST pass runs, no primary hit is shaded: what used to be 28 - 32 % of the kernel's cycles at a quarter of the lanes (EXPERIMENTS.md, round 4).
                typedef F FV4 __attribute__((ext_vector_type(4)));
                typedef typename FirstCode<F>::type Code;
                const FV4 *const recs = (const FV4 *)C.first;
                plist_count = 0xFFFFu;
                // (a record is 1 - 2 us away, in HBM, and a sample that ended at its first segment - 4 in 10 - only leads to the next record: the records of TWO samples are
                // asked for at once, so that a run of such samples costs half the round trips; what was fetched in vain warms the caches for the lane's next visit)
                auto take = [&](const FV4 &p0, const FV4 &p1, uint32_t code) -> bool { // -> the lane has a ray to follow
                    const uint32_t kind = code >> 30;
                    need_ray = false;
                    if (kind == kFirstDone) {
                        n_segments += 1; // (counted where the record is consumed: a sample the tail kernel finishes is traced there from its camera ray, and counted there)
                        finish_sample<F, RESUME, kPlain>(P, mk<F>(p0.x, p0.y, p0.z), task, s_cur, s_end, single, out_index, acc, need_task, need_ray);
                        return false;
                    }
                    if (kind == kFirstUnknown) { // traced from its camera ray, like any ray (its pixel has no usable list)
                        camera_ray<F>(C, pi, pj, s_cur, rng, path);
                        return true;
                    }
                    n_segments += 1;
                    const MaterialRec<F> m = P.mat[code & 0xFFFFu];
                    path.o = mk<F>(p0.x, p0.y, p0.z), path.d = mk<F>(p0.w, p1.x, p1.y);
                    path.tm = 0; // (scenes of spheres alone: nothing moves)
                    path.atten = m.type == 2 ? mk<F>((F)1.0, (F)1.0, (F)1.0) : mk<F>(m.r, m.g, m.b); // rrt.cu:58 at depth 0: (1, 1, 1) x albedo
                    path.depth = 1;
                    rng_open(rng, C.seed, (uint32_t)(pj * C.W + pi), (uint32_t)s_cur);
                    rng.n = (code >> 16) & 0x3FFFu;
                    return true;
                };
                while (need_ray) {
                    const uint32_t k = (uint32_t)(s_cur - sf);
                    const bool two = s_cur + 1 < s_end;
                    const uint32_t k2 = two ? k + 1u : k;
                    const FV4 p0 = recs[first_slot(task, k, (uint32_t)C.chunk, 0u)], p1 = recs[first_slot(task, k, (uint32_t)C.chunk, 1u)];
                    const FV4 q0 = recs[first_slot(task, k2, (uint32_t)C.chunk, 0u)], q1 = recs[first_slot(task, k2, (uint32_t)C.chunk, 1u)];
                    // (the code word through an integer load of its own: the compiler took element 0 for `bit_cast(p1.z)` of the vector load - ISA checked)
                    const uint32_t code = (uint32_t)((const Code *)(recs + first_slot(task, k, (uint32_t)C.chunk, 1u)))[2];
                    const uint32_t code2 = (uint32_t)((const Code *)(recs + first_slot(task, k2, (uint32_t)C.chunk, 1u)))[2];
                    if (take(p0, p1, code)) break;
                    if (need_ray && two && take(q0, q1, code2)) break; // (three and four at once: 35.3 / 36.1 ms against 34.8 for two and 35.7 for one)
                }
            }
        }

        // ---------------- pass mode --------------------------------------------------------------------
        // A camera ray (depth 0) can only hit the few spheres near its pixel's bundle of rays; those are
        // listed per pixel by primary_lists_kernel.  In a LIST pass the lanes holding a fresh camera ray
        // intersect it against their own short list (exact test) and shade, while the others wait; after
        // at most `list_passes` such passes (sky hits regenerate, so a lane may need several) a SCAN pass
        // runs the 488-sphere scan for every lane, by then almost all on secondary rays.  Every segment
        // is still intersected exactly once with the exact test in primitive order: same image.
        // (kDensePairs: which accelerated variants share the (ray, entry) pairs of a wave out over its lanes - see the other branch)
        if constexpr (!kDensePairs) {
        RRTX_SEC(2); // camera-ray lists (LIST passes) / walk
        // (a lane whose task ended among its first-bounce records has nothing to intersect: it waits for the next hand-out)
        const bool has_list = alive && !need_task && path.depth == 0 && plist_count != 0xFFFFu && P.max_depth > 0 && !(ACCEL != 0 && in_walk);
        const bool list_pass = list_passes_done < P.list_passes && __ballot(has_list) != 0ull;
        list_passes_done = list_pass ? list_passes_done + 1 : 0;

        // ---------------- LDSMODE == 3: the scan of this iteration, by the whole wave --------------------------------------------
        // The filter on the matrix cores (scenes of spheres alone; rrtx_pack.h: pack_mf_table has the operands and the bound):
        //   f = (c.n)^2 + b.c + g - thr  as ONE dot product of 31 f16 terms,
        // so two chained v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays leave the filter's VALUE in the result registers and the vector
        // unit the look at a sign per pair (the 7 FMAs of filter_value() are gone).  The instruction wants every lane of the wave - a
        // lane holds one sphere's terms, one ray's terms and the results of 16 spheres for ANOTHER lane's ray - so the scan runs HERE,
        // before the lanes part ways: a lane that does not scan in this iteration (not alive, no pass for it) hands in a column
        // nothing is a candidate for.  The lane that HOLDS a result marks the pair (ray, sphere); the marked pairs are then put to the
        // exact test with the ray's record from LDS, and hits go to the ray's owner through an LDS minimum over (t, index) - consider()'s
        // order-free rule, as in dense_candidates().  `best` is then what the scan section further down would have found.
        if constexpr (LDSMODE == 3) {
        const bool scans = alive && !need_task && P.max_depth > 0 && !list_pass;
        if (__ballot(scans) != 0ull) {
            RRTX_SEC(3);
            typedef _Float16 H8 __attribute__((ext_vector_type(8)));
            typedef float F4 __attribute__((ext_vector_type(4)));
            typedef F FV4 __attribute__((ext_vector_type(4)));
            // the wave's slice of cand_lds: [lists: 8 slots x 64 lanes x 32 bits][ray records: 64 x {o, a, d, -}], and before both the
            // 4 KB through which the rays' operands are transposed
            if (__builtin_expect(__builtin_amdgcn_read_exec() != ~0ull, 0)) atomicAdd(&P.counters[6], 1ull); // (the matrix instruction, the transposes and the queue below want all 64 lanes: rrtx_stats.convergence_faults)
            unsigned char *const area = (unsigned char *)&cand_lds[wave][0][0];
            U4 *const stage = (U4 *)area;
            uint32_t *const marks = (uint32_t *)area; // slot s of lane l at [s * 64 + l]: block << 16 | the signs of the lane's 16 results for it
            FV4 *const rays = (FV4 *)(area + 2048);   // ray r: {o, a} at [r], {d, -} at [64 + r] (two arrays: SQ_LDS_BANK_CONFLICT 8.5e8 -> 5.5e8 per C3 launch against one array of pairs, 47.85 -> 47.64 ms)
            const F a = vlen2<F>(path.d); // sphere.h:36
            best.t = Limits<F>::inf(), best.idx = -1;
            const FilterRay fr = make_filter_ray_mf(path, a);
            // (not sane - beyond what consider()'s order-free rule is proven for, accel_closest_hit has the same test, or so far out that the
            // scale below would leave the f16 range, |o|^2 >= 2^38: the lane scans by itself below, in the reference's order)
            const float fr_max = fmaxf(fmaxf(ffabs(fr.g), ffabs(fr.bx)), fmaxf(ffabs(fr.by), ffabs(fr.bz)));
            const bool sane = scans && fr.g < Limits<float>::inf() && fr_max < 0x1p38f && a >= Limits<F>::coop_tiny() && a <= Limits<F>::coop_big() && vlen2<F>(path.o) <= Limits<F>::coop_big();
            // the ray's 10 numbers, scaled by a power of two that brings the largest of them under 2^14 (f16 holds 65504; the scale itself, 2^-24 at the least, is one of the operands)
            float lam = 1.0f;
            {
                int e = 0;
                (void)__builtin_frexpf(sane ? fr_max : 1.0f, &e);
                lam = __builtin_ldexpf(1.0f, e > 14 ? 14 - e : 0);
            }
            const float rv[10] = {fr.nx * fr.nx * lam, fr.ny * fr.ny * lam, fr.nz * fr.nz * lam, fr.nx * fr.ny * lam, fr.nx * fr.nz * lam, fr.ny * fr.nz * lam,
                                  fr.bx * lam,         fr.by * lam,         fr.bz * lam,         fr.g * lam};
            _Float16 hh[10], hl[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const float x = sane ? rv[q] : 0.0f;
                hh[q] = (_Float16)x;
                hl[q] = (_Float16)(x - (float)hh[q]);
            }
            // A lane without a ray for this scan, or with one out of range, hands in the column nothing is a candidate for: g = -60000 twice
            // and thr x -1, so that f = -120000 - thr <= -60000 for a sphere of the table (|thr| <= 60000) and f = -60000 for a padding
            // record or a sphere listed apart (their only term is thr = 60000)
            if (!sane) hh[9] = hl[9] = (_Float16)(-60000.0f);
            const _Float16 ml = (_Float16)(sane ? -lam : -1.0f), z16 = (_Float16)0.0f;
            const H8 ch0 = {hh[0], hl[0], hh[0], hh[1], hl[1], hh[1], hh[2], hl[2]};
            const H8 ch1 = {hh[2], hh[3], hl[3], hh[3], hh[4], hl[4], hh[4], hh[5]};
            const H8 ch2 = {hl[5], hh[5], hh[6], hl[6], hh[6], hh[7], hl[7], hh[7]};
            const H8 ch3 = {hh[8], hl[8], hh[8], hh[9], hl[9], ml, ml, z16};
            // transpose through LDS: the instruction wants lane l to hold terms 16 kh + 8 (l / 32) .. + 7 of ray 32 h + l % 32 (swizzled against bank conflicts)
            {
                const uint32_t sw = ((uint32_t)lane >> 2) & 3u;
                stage[(uint32_t)lane * 4u + (0u ^ sw)] = __builtin_bit_cast(U4, ch0);
                stage[(uint32_t)lane * 4u + (1u ^ sw)] = __builtin_bit_cast(U4, ch1);
                stage[(uint32_t)lane * 4u + (2u ^ sw)] = __builtin_bit_cast(U4, ch2);
                stage[(uint32_t)lane * 4u + (3u ^ sw)] = __builtin_bit_cast(U4, ch3);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            H8 bt[2][2]; // [tile of 32 rays][half of the 32 terms]
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const uint32_t ray = 32u * (uint32_t)h + ((uint32_t)lane & 31u), chunk = 2u * (uint32_t)kh + ((uint32_t)lane >> 5);
                    bt[h][kh] = __builtin_bit_cast(H8, stage[ray * 4u + (chunk ^ ((ray >> 2) & 3u))]);
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // the rays' records for the exact tests, the owners' result slots
            {
                const FV4 r0 = {path.o.x, path.o.y, path.o.z, a}, r1 = {path.d.x, path.d.y, path.d.z, (F)0};
                rays[lane] = r0, rays[64 + lane] = r1; // (two arrays: a gather of 16 lanes meets 16 bank groups instead of 8)
                mf_keys[wave][lane] = ~0ull;
                if (sizeof(F) == 8) mf_ranks[wave][lane] = 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            typedef __attribute__((address_space(3))) uint32_t *LdsWords; // (a 32-bit pointer: the list's address arithmetic stays in one register)
            const LdsWords wp0 = (LdsWords)(marks + lane);
            LdsWords wp = wp0; // where this lane's next entry goes: slot s at marks[s * 64 + lane]
            // phase 2: the exact test (sphere.h:33-49) of the pairs whose filter value is not negative, each by the lane that holds the value.
            // An entry: block << 17 | tile h << 16 | 16 signs; bit 15 - v stands for the pair (ray 32 h + lane % 32, sphere
            // 32 block + 8 (v / 4) + 4 (lane / 32) + v % 4) - the instruction's result v of that tile; set = negative.
            // One pair to the exact test, its hit to the ray's owner: LDS minimum over (t, index) - consider()'s rule, as in dense_candidates()
            auto test_pair = [&](bool live, uint32_t owner, int idx) {
                const FV4 r0 = rays[owner], r1 = rays[64u + owner];
                const SphereHot<F> g = P.sph_hot[live ? idx : 0];
                const F ra = r0.w;
                const F ocx = r0.x - g.cx, ocy = r0.y - g.cy, ocz = r0.z - g.cz;
                const F half_b = ocx * r1.x + ocy * r1.y + ocz * r1.z;
                const F c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.r2;
                const F disc = half_b * half_b - ra * c;
                bool keep = live && !(disc < 0);
#if RRTX_SKIP_BEHIND
                keep = keep && !(half_b > 0 && c > 0); // (sphere_unordered has the argument)
#endif
                F t_hit = 0;
                if (keep) { // sphere.h:41-49 without the dependence on the scan order (resolve_pending)
                    const F sq = fsqrt(disc);
                    t_hit = (-half_b - sq) / ra;
                    if (t_hit < t_min) {
                        t_hit = (-half_b + sq) / ra;
                        keep = !(t_hit < t_min);
                    }
                }
                if (sizeof(F) == 4) {
                    if (keep) atomicMin(&mf_keys[wave][owner], ((unsigned long long)__float_as_uint((float)t_hit) << 32) | (unsigned long long)(~(uint32_t)idx)); // ds_min_u64: smallest t, then largest index
                }
                else { // (a sequence of steps the lanes take together)
                    const unsigned long long tb = (unsigned long long)__double_as_longlong((double)t_hit);
                    const unsigned long long before = mf_keys[wave][owner];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (keep) atomicMin(&mf_keys[wave][owner], tb);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const bool wins = keep && mf_keys[wave][owner] == tb; // this pair's t is the owner's minimum so far
                    if (wins && tb < before) mf_ranks[wave][owner] = 0u;  // ... a new one: what the slot says belongs to a larger t
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (wins) atomicMax(&mf_ranks[wave][owner], (uint32_t)idx + 1u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            };
            // A lane's entries decoded pair by pair (every lane takes every trip): f(live, owner, sphere)
            auto for_each_pair = [&](uint32_t cnt, auto &&f) {
                uint32_t k = 0, bits = 0, blk = 0, tile = 0; // (blk: the block's first sphere)
                for (;;) {
                    if (bits == 0u && k < cnt) {
                        const uint32_t e = marks[k * 64u + (uint32_t)lane];
                        bits = ~e & 0xFFFFu, blk = (e >> 17) << 5, tile = (e >> 16) & 1u, k += 1;
                    }
                    if (__ballot(bits != 0u) == 0ull) break;
                    const bool live = bits != 0u;
                    const uint32_t q = live ? 15u - (uint32_t)__builtin_ctz(bits) : 0u;
                    bits &= bits - 1u;
                    f(live, 32u * tile + ((uint32_t)lane & 31u), (int)(blk + 8u * (q >> 2) + 4u * ((uint32_t)lane >> 5) + (q & 3u)));
                }
            };
            // (in mid-scan, when a lane's list is full - practically never: every lane tests what it listed, the scan's registers stay where they are)
            auto drain_in_scan = [&]() {
                RRTX_SEC(4);
                for_each_pair((uint32_t)(wp - wp0) >> 6, [&](bool live, uint32_t owner, int idx) {
                    n_candidates += live ? 1u : 0u;
                    test_pair(live, owner, idx);
                });
                wp = wp0;
                RRTX_SEC(3);
            };
            auto drain_mf = [&]() {
                RRTX_SEC(4);
                const uint32_t cnt = (uint32_t)(wp - wp0) >> 6;
#if RRTX_MF_BALANCE
                // The pairs are few (1.8 a ray) and unevenly held (the busiest lane of 64: 6): they are laid end to end in a queue (a prefix
                // sum over the lanes' counts) and tested 128 at a time, two to a lane, whoever listed them - the two tests' loads and roots in flight together.
                uint32_t np = 0;
                for (uint32_t k = 0; __ballot(k < cnt) != 0ull; ++k) np += k < cnt ? (uint32_t)__builtin_popcount(~marks[k * 64u + (uint32_t)lane] & 0xFFFFu) : 0u;
                n_candidates += np;
                const uint32_t incl = wave_scan_add(np), total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (total <= (uint32_t)kMfQueue) {
                    uint32_t pos = incl - np;
                    for_each_pair(cnt, [&](bool live, uint32_t owner, int idx) {
                        if (live) mf_queue[wave][pos] = (uint16_t)((owner << 10) | (uint32_t)idx);
                        pos += live ? 1u : 0u;
                    });
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (uint32_t base = 0; base < total; base += 128u) {
                        const bool live0 = base + (uint32_t)lane < total, live1 = base + 64u + (uint32_t)lane < total;
                        const uint32_t e0 = live0 ? (uint32_t)mf_queue[wave][base + (uint32_t)lane] : 0u, e1 = live1 ? (uint32_t)mf_queue[wave][(base + 64u + (uint32_t)lane) & (uint32_t)(kMfQueue - 1)] : 0u;
                        test_pair(live0, e0 >> 10, (int)(e0 & 1023u));
                        if (__ballot(live1) != 0ull) test_pair(live1, e1 >> 10, (int)(e1 & 1023u));
                    }
                }
                else
                    for_each_pair(cnt, test_pair);
#else
                for_each_pair(cnt, [&](bool live, uint32_t owner, int idx) {
                    n_candidates += live ? 1u : 0u;
                    test_pair(live, owner, idx);
                });
#endif
                wp = wp0;
                RRTX_SEC(3);
            };
            // the spheres the f16 table cannot hold (the r = 1000 ground sphere): exact, by the owner, wave-uniform index
            {
                const RRTX_CONST_AS uint32_t *big = (const RRTX_CONST_AS uint32_t *)P.mf_big;
                PendingRoot<F> pend = {-1, 0, 0};
                for (int i = 0; i < P.n_mf_big; ++i) {
                    const int idx = (int)big[i];
                    const SphereHot<F> g = P.sph_hot[idx];
                    if (sane) sphere_unordered<F>(g.cx, g.cy, g.cz, g.r2, path, a, t_min, idx, kNoTriangles, best, pend);
                }
                resolve_pending<F>(pend, a, t_min, kNoTriangles, best);
                if (sane) n_candidates += (uint32_t)P.n_mf_big;
            }
            // phase 1, without a branch: a tile's products (32 spheres x 32 rays, two chained v_mfma_f32_32x32x16_f16) are issued before the
            // previous tile's results are looked at; what is looked at is their SIGN (v_alignbit shifts it into a mask, an instruction
            // per pair).  A sign is a verdict because a sum is never -0 here: the three terms Q_ii,hi x N_ii,hi are squares times squares,
            // +0 at the least, and round-to-nearest adds +0 and -0 to +0 (sphere.h:41 reads !(disc < 0): -0 would be a candidate carrying
            // the sign of none); NaN does not arise from finite f16.
            typedef float F16v __attribute__((ext_vector_type(16)));
            const F16v zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto signs16 = [](const F16v &x) {
                uint32_t ma = 0, mb = 0;
#pragma unroll
                for (int v = 0; v < 8; ++v) ma = __builtin_amdgcn_alignbit(ma, __float_as_uint(x[v]), 31), mb = __builtin_amdgcn_alignbit(mb, __float_as_uint(x[8 + v]), 31);
                return (ma << 8) | mb;
            };
            const int n_blocks = (n_sph_pad + 31) >> 5;
            auto table_block = [&](int b, int kh) { return __builtin_bit_cast(H8, mf_lds[(uint32_t)((b < n_blocks ? b : n_blocks - 1) * 2 + kh) * 64u + (uint32_t)lane]); }; // (past the end: the last block once more, nobody looks)
            auto products = [&](const H8 &a0, const H8 &a1, int h) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bt[h][1], __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bt[h][0], zero16, 0, 0, 0), 0, 0, 0); };
            auto look = [&](const F16v &f, int b, int h) {
                const uint32_t m = signs16(f);
                if (m != 0xFFFFu) {
                    *wp = ((uint32_t)(2 * b + h) << 16) | m;
                    wp += 64;
                }
            };
            const LdsWords wp_full = wp0 + 64 * (kMfSlots - 2); // a block lists up to two entries a lane
#if RRTX_MF_PRIO
            __builtin_amdgcn_s_setprio(RRTX_MF_PRIO > 0 ? RRTX_MF_PRIO : 0);
#endif
            H8 a0 = table_block(0, 0), a1 = table_block(0, 1);
            F16v f0 = products(a0, a1, 0), f1;
            for (int b = 0; b < n_blocks; ++b) {
                f1 = products(a0, a1, 1);
                a0 = table_block(b + 1, 0), a1 = table_block(b + 1, 1);
                if (__ballot(wp > wp_full) != 0ull) drain_in_scan();
                look(f0, b, 0);
                f0 = products(a0, a1, 0);
                look(f1, b, 1);
            }
#if RRTX_MF_PRIO
            __builtin_amdgcn_s_setprio(RRTX_MF_PRIO > 0 ? 0 : -(RRTX_MF_PRIO));
#endif
            drain_mf();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            {
                const unsigned long long k = mf_keys[wave][lane];
                if (k != ~0ull) {
                    if (sizeof(F) == 4)
                        consider<F>((F)__uint_as_float((uint32_t)(k >> 32)), (int)~(uint32_t)k, kNoTriangles, best);
                    else
                        consider<F>((F)__longlong_as_double((long long)k), (int)(mf_ranks[wave][lane] - 1u), kNoTriangles, best);
                }
            }
            if (__builtin_expect(scans && !sane, 0)) best = sequential_closest_hit<F>(P, path, a, t_min); // hittable_list.h:95-117, as it stands
            RRTX_SEC(2);
        }
        }
        if (alive && !need_task && (!list_pass || has_list)) {
            bool done = false;
            V3<F> radiance = mk<F>(0, 0, 0);
            if (P.max_depth <= 0) { // rrt.cu:47 loop body never runs
                done = true;
            }
            else {
            // ---------------- closest hit: hittable_list.h:95-117 ---------------------------------
            if (!(ACCEL != 0 && in_walk)) n_segments += 1;
            const F a = vlen2<F>(path.d); // sphere.h:36
            if (!(ACCEL != 0 && in_walk) && !(LDSMODE == 3 && !list_pass)) {
                best.t = Limits<F>::inf();
                best.idx = -1;
            }
            bool still_walking = false;
            if (list_pass) {
                // exact tests in primitive order: listed spheres, then every moving sphere and triangle
                // (the pixel's 32-byte list comes in with two 16-byte loads and is parked in this lane's candidate slots in LDS -
                // free during a LIST pass -, the entries are picked from there instead of one by one from L2, each a dependent load
                // in front of the record's; use_bvh 41.3 -> 41.0 ms, list scan unchanged: the LIST passes - 22.7 % of the accelerated
                // kernel's cycles, 9.2 % of the list scan's, a quarter of the lanes at work - are bound by issue, not by these loads)
                static_assert(kPlistStride == 16 && kCap >= 8, "a pixel's list is 8 dwords");
                const auto &C = *cold_params<F>();
                typedef uint32_t U4 __attribute__((ext_vector_type(4)));
                const U4 *pl4 = (const U4 *)(C.plist + (size_t)task_pixel<F, kPlain>(C, task) * kPlistStride);
                const U4 lo4 = pl4[0], hi4 = pl4[1];
                my_cand[0 * 64] = lo4.x, my_cand[1 * 64] = lo4.y, my_cand[2 * 64] = lo4.z, my_cand[3 * 64] = lo4.w;
                my_cand[4 * 64] = hi4.x, my_cand[5 * 64] = hi4.y, my_cand[6 * 64] = hi4.z, my_cand[7 * 64] = hi4.w;
                n_candidates += plist_count;
                for (uint32_t k = 1; k <= plist_count; ++k) { // halfword 0 is the count, entries follow
                    const int idx = (int)((my_cand[(k >> 1) * 64] >> ((k & 1u) * 16u)) & 0xFFFFu);
                    const SphereHot<F> gq = ACCEL == 2 ? hot_lds[idx] : P.sph_hot[idx];
                    refine_sphere<F>(gq.cx, gq.cy, gq.cz, gq.r2, path, a, t_min, idx, best);
                }
                for (int q = 0; q < n_msph; ++q) {
                    const MovingSphereRec<F> ms = P.msph[q];
                    const V3<F> cen = msphere_center<F>(ms, path.tm);
                    refine_sphere<F>(cen.x, cen.y, cen.z, ms.r2, path, a, t_min, msph_base + q, best);
                }
                for (int q = 0; q < n_tri; ++q) {
                    F tt;
                    if (triangle_test<F, true>(P.tri[q], path, t_min, best.t, tt)) {
                        best.t = tt;
                        best.idx = tri_base + q;
                    }
                }
                if (VERIFY) { // test build of the kernel: the list must reproduce the full sequential scan
                    const HitInfo<F> full = sequential_closest_hit<F>(P, path, a, t_min);
                    if (full.idx != best.idx || !(full.t == best.t)) atomicAdd(&P.counters[2], 1ull);
                }
            }
            else {
            bool need_scan = true;
            if (ACCEL != 0) RRTX_SEC(7); // (developer builds: the grid walk, apart from the LIST passes of section 2)
            if (ACCEL != 0) {
                const auto &C = *cold_params<F>(); // the grid's geometry is wanted here only
                // (tables in LDS: 4 cells, a constant the loop is compiled for - 37.8 -> 37.5 ms -; the grids that ask for 16, large and
                // mostly empty, live in HBM; the macro: experiments)
                const int slice = RRTX_WALK_SLICE > 0 ? RRTX_WALK_SLICE : (ACCEL == 2 ? 4 : C.grid.walk_slice);
                int r;
                if (in_walk && walk_cell == kCoopDone)
                    r = kWalkDone; // a far ray the wave resolved at the top of this iteration: `best` is final
                else if (ACCEL == 2)
                    r = accel_closest_hit<F, SO>(C, hot_lds, cell_start_lds, cell_prims_lds, path, a, t_min, best, in_walk, walk_cell, walk_t_out, slice);
                else
                    r = accel_closest_hit<F, SO>(C, P.sph_hot, P.grid_cell_start, P.grid_cell_prims, path, a, t_min, best, in_walk, walk_cell, walk_t_out, slice);
                need_scan = r == kWalkNeedsScan;
                if (r == kWalkFarScan) walk_cell = kCoopWait; // (waits, as a walk in progress, for the next top of the loop)
                still_walking = in_walk = r == kWalkGoesOn || r == kWalkFarScan;
                if (VERIFY && !need_scan && !still_walking) { // test build of the kernel: the walk must reproduce the full sequential scan
                    const HitInfo<F> full = sequential_closest_hit<F>(P, path, a, t_min);
                    if (full.idx != best.idx || !(full.t == best.t)) atomicAdd(&P.counters[2], 1ull);
                }
            }
            if (need_scan) {
            RRTX_SEC(3); // scan phase 1 (filter + pushes; the drains in between go to section 4)
            n_scanned += 1;
            if constexpr (LDSMODE == 3) {
                // (the scan has run already, by the whole wave, before the lanes parted: `best` holds its result)
            }
            else {
            uint32_t cnt = 0;

            // phase 2 body, used for flushes and at the end
            auto drain = [&]() {
                RRTX_SEC(4); // phase 2: exact refinement of the candidates
                n_candidates += cnt;
                for (uint32_t s = 0; s < cnt; ++s) {
                    const int k = (int)my_cand[s * 64];
                    if (k < msph_base) {
                        if (k < n_sph) {
                            const SphereHot<F> g = P.sph_hot[k];
                            refine_sphere<F>(g.cx, g.cy, g.cz, g.r2, path, a, t_min, k, best);
                        }
                    }
                    else if (k < tri_base) {
                        const MovingSphereRec<F> m = P.msph[k - msph_base];
                        const V3<F> cen = msphere_center<F>(m, path.tm);
                        refine_sphere<F>(cen.x, cen.y, cen.z, m.r2, path, a, t_min, k, best);
                    }
                    else {
                        F t;
                        if (triangle_test<F, true>(P.tri[k - tri_base], path, t_min, best.t, t)) {
                            best.t = t;
                            best.idx = k;
                        }
                    }
                }
                cnt = 0;
                RRTX_SEC(3);
            };

            // phase 1a: spheres, wave-uniform scalar loads.  The block of kUnroll tests is kept
            // branch-free (all tests first, pushes afterwards) so that the scalar loads of a whole
            // block are issued together and the VALU stream is one straight line.
            //
            // FILTER = false: the reference's discriminant itself (sphere.h:35-40), 17 VALU + 1 compare.
            // FILTER = true : a CONSERVATIVE test in 7 FMAs + 1 compare that can only err towards
            //   "candidate" (DESIGN.md "Conservative scan filter" has the bound).  With n = d/|d|,
            //   u = c.n, s = o.n:   disc/|d|^2 = u^2 + 2(o - s n).c + (s^2 - |o|^2) + (r^2 - |c|^2).
            //   Per segment: n, b = 2(o - s n), g = s^2 - |o|^2 + K eps |o|^2.  Per sphere (SGPRs):
            //   c and thr = |c|^2 - r^2 - K eps (|c|^2 + r^2), rounded down on the host.
            //   candidate  <=>  not (u^2 + b.c + g < thr).   Phase 2 then applies the exact test.
            FilterRay fr = {0, 0, 0, 0, 0, 0, Limits<float>::inf()};
            if (FILTER) fr = make_filter_ray(path, a); // extreme rays take the always-candidate route: phase 2 is exact
            constexpr int kUnroll = SphereUnroll<ST>::value;
            // one straight-line block of N tests; FROM_LDS selects the operand source
            auto scan_block = [&](int k0, auto from_lds_c, auto n_c) {
                constexpr bool FROM_LDS = decltype(from_lds_c)::value != 0;
                constexpr int N = decltype(n_c)::value;
                if (__ballot(cnt > (uint32_t)(kCap - N)) != 0ull) drain();
                // the block's records first (s_load_dwordx16s, or N ds_read_b128 broadcasts), then
                // N x {test, push}
                ST bcx[N], bcy[N], bcz[N], bw[N];
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    if (FROM_LDS) {
                        const SphereHot<ST> r = sph_lds[k0 + u];
                        bcx[u] = r.cx, bcy[u] = r.cy, bcz[u] = r.cz, bw[u] = r.r2;
                    }
                    else {
                        bcx[u] = sph_scalar[k0 + u].cx, bcy[u] = sph_scalar[k0 + u].cy, bcz[u] = sph_scalar[k0 + u].cz, bw[u] = sph_scalar[k0 + u].r2;
                    }
                }
                auto test = [&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const ST cx = bcx[u], cy = bcy[u], cz = bcz[u], r2 = bw[u];
                    if (FILTER) {
                        push_if_not_less<u, FROM_LDS>(filter_value(fr, (float)cx, (float)cy, (float)cz), (float)r2, cnt, my_cand_lds, my_cand, k0); // the r2 slot holds thr
                    }
                    else {
                        const F ocx = path.o.x - (F)cx, ocy = path.o.y - (F)cy, ocz = path.o.z - (F)cz;
                        const F half_b = ocx * path.d.x + ocy * path.d.y + ocz * path.d.z;
                        const F c = (ocx * ocx + ocy * ocy + ocz * ocz) - (F)r2;
                        const F disc = half_b * half_b - a * c;
                        push_if_not_less<u, false>(disc, (F)zero_sgpr, cnt, my_cand_lds, my_cand, k0); // !(disc < 0), sphere.h:41
                    }
                };
                test(IntC<0>());
                test(IntC<1 % N>());
                if (N > 2) {
                    test(IntC<2 % N>());
                    test(IntC<3 % N>());
                }
                if (N > 4) {
                    test(IntC<4 % N>());
                    test(IntC<5 % N>());
                    test(IntC<6 % N>());
                    test(IntC<7 % N>());
                }
            };
            // LDS blocks are half as long as scalar blocks: half the vector registers for the same bytes
            constexpr int kLdsN = RRTX_LDS_BLOCK < kUnroll ? RRTX_LDS_BLOCK : kUnroll;
            for (int k0 = 0; k0 < n_sph_pad; k0 += 2 * kUnroll) { // the tables are padded to 2 * kUnroll records
                if (LDSMODE == 0) {
                    scan_block(k0, IntC<0>(), IntC<kUnroll>());
                    scan_block(k0 + kUnroll, IntC<0>(), IntC<kUnroll>());
                }
                else {
                    if (LDSMODE == 2) {
#pragma unroll
                        for (int j = 0; j < kUnroll; j += kLdsN) scan_block(k0 + j, IntC<1>(), IntC<kLdsN>());
                    }
                    else
                        scan_block(k0, IntC<0>(), IntC<kUnroll>());
#pragma unroll
                    for (int j = 0; j < kUnroll; j += kLdsN) scan_block(k0 + kUnroll + j, IntC<1>(), IntC<kLdsN>());
                }
            }
            // phase 1b: moving spheres (center depends on the ray's time: per-lane)
            const RRTX_CONST_AS MovingSphereRec<F> *msph_scalar = (const RRTX_CONST_AS MovingSphereRec<F> *)P.msph; // (uniform index: s_load)
            for (int m = 0; m < n_msph; ++m) {
                if (__ballot(cnt >= (uint32_t)kCap) != 0ull) drain();
                const RRTX_CONST_AS MovingSphereRec<F> &ms = msph_scalar[m];
                const V3<F> cen = msphere_center<F>(ms, path.tm);
                const F ocx = path.o.x - cen.x, ocy = path.o.y - cen.y, ocz = path.o.z - cen.z;
                const F half_b = ocx * path.d.x + ocy * path.d.y + ocz * path.d.z;
                const F c = (ocx * ocx + ocy * ocy + ocz * ocz) - ms.r2;
                const F disc = half_b * half_b - a * c;
                if (!(disc < 0)) {
                    my_cand[cnt * 64] = (uint32_t)(msph_base + m);
                    cnt += 1;
                }
            }
            // phase 1c: triangles
            // (one aligned s_load_dwordx16 per triangle through the constant address space, see TriScanRec)
            const RRTX_CONST_AS TriScanRec<F> *tri_scalar = (const RRTX_CONST_AS TriScanRec<F> *)P.tri_scan;
            for (int t = 0; t < n_tri; ++t) {
                if (__ballot(cnt >= (uint32_t)kCap) != 0ull) drain();
                // (as one vector: the compiler otherwise loads field by field, where each is first used; loading
                // one triangle ahead gained nothing: 362 vs 355 ms)
                typedef F Vec16 __attribute__((ext_vector_type(16)));
                const Vec16 q = *(const RRTX_CONST_AS Vec16 *)&tri_scalar[t];
                TriScanRec<F> rec;
                rec.v0[0] = q[0], rec.v0[1] = q[1], rec.v0[2] = q[2], rec.e1[0] = q[3], rec.e1[1] = q[4], rec.e1[2] = q[5], rec.e2[0] = q[6], rec.e2[1] = q[7], rec.e2[2] = q[8];
                F dummy;
                if (triangle_test<F, false>(rec, path, t_min, best.t, dummy)) {
                    my_cand[cnt * 64] = (uint32_t)(tri_base + t);
                    cnt += 1;
                }
            }
            drain();
            } // LDSMODE != 3
            } // need_scan
            } // scan pass

            // ---------------- shade: rrt.cu:49-76 -------------------------------------------------
            RRTX_SEC(5); // shading
            if (!still_walking) done = shade<F, SO>(P, best, path, rng, radiance);
            } // max_depth > 0

            RRTX_SEC(6); // sample / task bookkeeping, stores
            if (done) finish_sample<F, RESUME, kPlain>(P, radiance, task, s_cur, s_end, single, out_index, acc, need_task, need_ray);
        }
        } // !kDensePairs
        else {
            // ---------------- accelerated closest hit: the wave's (ray, entry) pairs tested densely ------------------
            // Every lane lists what it wants tested in this iteration: a fresh camera ray its pixel's candidate list (and every
            // moving sphere and triangle), any other segment the entries of the next cells of its walk through the grid
            // (accel_walk_prepare: the always-list and the box clip of a new segment, then up to `slice` cells of the DDA); the 64
            // lanes then test all those pairs together (dense_candidates), every owner finishes its few candidates with the
            // exact test and consider()'s order-independent rule, and walks whose slice settled the closest hit go on to shade.
            // There are no LIST and SCAN passes here: camera rays and walks are served by the same trips.
            RRTX_SEC(2);
            bool done = false, resolved = false, need_scan = false;
            V3<F> radiance = mk<F>(0, 0, 0);
            const auto &C = *cold_params<F>(); // the grid's geometry is wanted here only
            // (tables in LDS: slices of 4 cells; the grids that ask for more, large and mostly empty, live in HBM; the macro: experiments)
            const int slice = RRTX_WALK_SLICE > 0 ? RRTX_WALK_SLICE : (ACCEL == 2 ? 4 : C.grid.walk_slice);
            WalkRanges R = {0u, 0u, 0u, 0u, 0u, 0u};
            bool is_list = false, walking = false, ended = false;
            F t_last = 0, slack_t = 0, a = 0;
            const bool has_ray = alive && !need_task; // (a lane whose task ended among its first-bounce records waits for the next hand-out)
            if (has_ray && P.max_depth <= 0) // rrt.cu:47 loop body never runs
                done = true;
            else if (has_ray) {
                a = vlen2<F>(path.d); // sphere.h:36
                if (!in_walk) {
                    n_segments += 1;
                    best.t = Limits<F>::inf();
                    best.idx = -1;
                }
                if (in_walk && walk_cell == kCoopDone) // a far ray the wave resolved at the top of this iteration: `best` is final
                    resolved = true, in_walk = false;
                else if (!in_walk && path.depth == 0 && plist_count != 0xFFFFu) {
                    // a fresh camera ray with a list.  The unordered rule needs a sane ray (the scene is finite wherever a grid was built)
                    const F o2 = path.o.x * path.o.x + path.o.y * path.o.y + path.o.z * path.o.z;
                    if (a >= Limits<F>::coop_tiny() && a <= Limits<F>::coop_big() && o2 <= Limits<F>::coop_big() && ffabs(path.tm) <= Limits<F>::coop_big()) {
                        is_list = true;
                        R.beg0 = task_pixel<F, kPlain>(C, task), R.cnt = plist_count;
                        if (!SO) R.beg1 = (uint32_t)msph_base, R.beg2 = (uint32_t)tri_base, R.cnt |= ((uint32_t)n_msph << 8) | ((uint32_t)n_tri << 16); // (<= 64 of them where lists exist)
                    }
                    else
                        need_scan = true;
                }
                else {
                    int r;
                    // (empty-space skipping - an occupancy byte per block of 4 x 4 x 4 cells, empty blocks crossed in one step: accel_walk_prepare, checked
                    // on the host - measured on the 27 072-triangle mesh: 16.0 -> 20.1 ms with the bytes read from HBM (fp64 21.9 -> 23.9), 21.7 ms from a copy
                    // in LDS: the jump - three boundary distances, the landing cell, the DDA's state rebuilt - costs more than the 2 - 3 empty cells it
                    // saves.  Not used: EXPERIMENTS.md)
                    const uint8_t *coarse = nullptr;
                    if (ACCEL == 2)
                        r = accel_walk_prepare<F, SO>(C, hot_lds, cell_start_lds, cell_prims_lds, path, a, t_min, best, in_walk, walk_cell, walk_t_out, slice, R, t_last, slack_t, ended, coarse);
                    else
                        r = accel_walk_prepare<F, SO>(C, P.sph_hot, P.grid_cell_start, P.grid_cell_prims, path, a, t_min, best, in_walk, walk_cell, walk_t_out, slice, R, t_last, slack_t, ended, coarse);
                    if (r == kWalkDone)
                        resolved = true, in_walk = false;
                    else if (r == kWalkNeedsScan)
                        need_scan = true, in_walk = false;
                    else if (r == kWalkFarScan)
                        walk_cell = kCoopWait, in_walk = true; // (waits, as a walk in progress, for the next top of the loop)
                    else
                        walking = true;
                }
            }
            // ---- all 64 lanes: the pairs of the wave
            RRTX_SEC(7);
            const uint32_t n_own = walk_range_total(R.cnt);
            n_walk_cells += R.steps, n_walk_pairs += n_own;
            if (ACCEL == 2)
                dense_candidates<F, SO, kCap>(P, hot_lds, cell_prims_lds, C.plist, path, a, R, t_min, is_list, n_own, lane, &dense_marks[wave][0], &dense_ranks[wave][0], &dense_keys[wave][0]);
            else
                dense_candidates<F, SO, kCap>(P, P.sph_hot, P.grid_cell_prims, C.plist, path, a, R, t_min, is_list, n_own, lane, &dense_marks[wave][0], &dense_ranks[wave][0], &dense_keys[wave][0]);
            // ---- every owner: its hits, in any order
            RRTX_SEC(3);
            if (n_own != 0u) {
                const unsigned long long k = dense_keys[wave][lane];
                if (k != ~0ull) {
                    F t_hit;
                    uint32_t rank_u;
                    if (sizeof(F) == 4)
                        t_hit = (F)__uint_as_float((uint32_t)(k >> 32)), rank_u = ~(uint32_t)k;
                    else
                        t_hit = (F)__longlong_as_double((long long)k), rank_u = dense_ranks[wave][lane] - 1u;
                    const int r = (int)(rank_u ^ 0x80000000u);
                    consider<F>(t_hit, r >= 0 ? r : -r - 1, SO ? kNoTriangles : tri_base, best);
                    n_candidates += 1;
                }
            }
#ifdef RRTX_SECTION_DIAG // developer builds: what the dense pairing is fed, per wave and iteration -> counters[8 + k]
            {
                uint32_t pairs_dbg = n_own;
                for (int off_dbg = 32; off_dbg >= 1; off_dbg >>= 1) pairs_dbg += __shfl_xor(pairs_dbg, off_dbg);
                dense_dbg[0] += 1, dense_dbg[1] += (pairs_dbg + 63u) / 64u, dense_dbg[2] += pairs_dbg;
                dense_dbg[6] += (unsigned long long)__popcll(__ballot(walking)), dense_dbg[7] += (unsigned long long)__popcll(__ballot(is_list));
            }
#endif
            RRTX_SEC(4);
            if (is_list) resolved = true;
            if (walking) {
                in_walk = accel_walk_decide<F>(best, t_last, slack_t, ended) == kWalkGoesOn;
                resolved = !in_walk;
                // (Every walk ends: a slice takes up at least its first cell - accel_walk_prepare: a cell of more entries than four ranges hold is
                // tested on the spot, any other fits the four empty ranges a slice starts with - and the DDA moves one cell along one axis in a
                // fixed direction per step, NaN or infinite boundary distances included (the comparisons then pick z), so it leaves the grid
                // after at most dims[0] + dims[1] + dims[2] steps; tests/path_host_check.cpp runs the same source on 2 M hostile rays.  Round 3
                // carried a counter here that sent a walk of more slices than that to the reference's scan; it never fired - round 4 removed it.)
            }
            if (VERIFY && resolved) { // test build of the kernel: lists and walks must reproduce the full sequential scan
                const HitInfo<F> full = sequential_closest_hit<F>(P, path, a, t_min);
                if (full.idx != best.idx || !(full.t == best.t)) atomicAdd(&P.counters[2], 1ull);
            }
            if (__builtin_expect(need_scan, 0)) { // a ray the unordered rule is not proven for: the reference's scan, in its order
                best = sequential_closest_hit<F>(P, path, a, t_min);
                n_scanned += 1;
                resolved = true;
            }
            // ---------------- shade: rrt.cu:49-76 -------------------------------------------------
            RRTX_SEC(5);
            if (resolved) done = shade<F, SO>(P, best, path, rng, radiance);
            RRTX_SEC(6); // sample / task bookkeeping, stores
            if (done) finish_sample<F, RESUME, kPlain>(P, radiance, task, s_cur, s_end, single, out_index, acc, need_task, need_ray);
        }
    }

    if (P.collect_stats) {
        atomicAdd(&P.counters[0], (unsigned long long)n_segments);
        atomicAdd(&P.counters[1], (unsigned long long)n_candidates);
        atomicAdd(&P.counters[3], (unsigned long long)n_scanned);
        if (kDensePairs && n_walk_cells) atomicAdd(&P.counters[4], (unsigned long long)n_walk_cells), atomicAdd(&P.counters[5], (unsigned long long)n_walk_pairs);
    }
#ifdef RRTX_RESUME_DIAG // developer builds (tools/resume_diag.py): the longest wave of a resume pass - iterations, clock cycles -, and the pass's totals
    if (RESUME) {
        if (lane == 0) {
            atomicMax(&P.counters[24], (unsigned long long)loop_count);
            atomicMax(&P.counters[25], (unsigned long long)(__builtin_amdgcn_s_memtime() - resume_t0));
            atomicAdd(&P.counters[26], (unsigned long long)loop_count);
            atomicAdd(&P.counters[27], loop_count > 1 ? 1ull : 0ull);
        }
        if (n_segments) atomicAdd(&P.counters[28], (unsigned long long)n_segments);
    }
#endif
#ifdef RRTX_SECTION_DIAG
    RRTX_SEC(7);
#ifdef RRTX_SECTION_RESUME // (the resume pass's sections instead of the render pass's: tools/resume_sections.py)
    if (lane == 0 && RESUME)
#else
    if (lane == 0 && !RESUME)
#endif
        for (int k = 0; k < 8; ++k) atomicAdd(&P.counters[16 + k], sec_cycles[k]), atomicAdd(&P.counters[8 + k], dense_dbg[k]);
#endif
#undef RRTX_SEC
#ifdef RRTX_DIAG
    if (lane == 0 && !RESUME) {
        const uint32_t wid = (blockIdx.x * kBT + threadIdx.x) >> 6;
        unsigned long long *d = P.diag + (size_t)wid * 8;
        d[0] = diag_t0, d[1] = diag_dry, d[2] = __builtin_amdgcn_s_memrealtime(), d[3] = diag_iters, d[4] = diag_iters_dry;
        d[5] = diag_pull_t, d[6] = ((unsigned long long)diag_pulls << 32) | diag_pull_iter, d[7] = diag_pull_base;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Tail kernel: finishes the work items the render kernel parked (see "hand-off" there).
//
// The parked population is what a launch ends with: a few hundred thousand paths, most of them long
// (paths are in flight for as long as they are long), each a chain of dependent segments.  G lanes
// share one ray here (64 / G rays per wave): every lane of a group carries the same path state, so
// shading and RNG run in lock step, and the scan is turned around: lane `sub` tests primitives sub,
// sub + G, ... with the EXACT reference test, then a butterfly inside the group picks the winner with
// the sequential scan's tie rules — equal t: the LAST sphere-like primitive wins (sphere.h:46-48
// accepts root == t_max); a triangle never displaces an equal t (triangle.h:63), so among triangles
// the FIRST wins and any sphere beats it.  Same arithmetic per primitive as the lane-per-ray scan,
// hence the same bits.
//
// Work units (written by the render kernel next to the items, P.tail_units): unit 0 of an item
// continues its partial sum over all but the last kTailSplit - 1 remaining samples of the task, units
// k >= 1 are those last samples one by one; tail_sum_kernel adds the results in sample order.  That
// bounds most units to one path.  Units are pulled by the wave in guided batches (large while many
// remain, one per group at the end) and dealt to the groups as they finish, as the render kernel deals
// tasks to lanes.
// ---------------------------------------------------------------------------------------------
template <typename F, int G, bool LDS, bool FILTER> __global__ void __launch_bounds__(kBlockThreads) tail_kernel(const KernelParams<F> P)
{
    // the sphere table is read from a copy in LDS when it fits (the scan is a chain of dependent loads
    // otherwise: ~1 us each from L2 under load)
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    typedef typename ScanType<F, FILTER>::type ST;
    const SphereHot<ST> *sph_tab = ScanType<F, FILTER>::table(P); // what every lane tests its share against
    const SphereHot<F> *exact_tab = P.sph_hot;                      // FILTER: the exact-test records of the candidates
    if (LDS) {
        SphereHot<ST> *const copy = (SphereHot<ST> *)dyn_lds;
        SphereHot<F> *const exact_copy = (SphereHot<F> *)(copy + P.n_sph_padded);
        for (int i = threadIdx.x; i < P.n_sph_padded; i += kBlockThreads) {
            copy[i] = sph_tab[i];
            if (FILTER) exact_copy[i] = P.sph_hot[i];
        }
        __syncthreads();
        sph_tab = copy;
        if (FILTER) exact_tab = exact_copy;
    }
    constexpr uint32_t kGroups = 64 / G;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const uint64_t lanes_below_group = (1ull << (lane & ~(G - 1))) - 1ull;
    const uint32_t n_waves = gridDim.x * (uint32_t)kWavesPerBlock;
    const uint32_t n_units = P.tail_count[2];
#ifdef RRTX_DIAG
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const F t_min = (F)0.001;
    const int n_msph = P.n_msph, n_tri = P.n_tri;
    const int n_sph_pad = P.n_sph_padded, msph_base = P.n_sph_padded, tri_base = P.n_sph_padded + n_msph;
    uint32_t n_segments = 0;

    // wave-uniform pool of units
    uint32_t pool_next = 0, pool_end = 0, last_base = 0;
    bool dry = false;
    // group state (identical in the G lanes of a group)
    bool have = false, retired = false, need_ray = false, single = false;
    uint32_t task = 0, out_index = 0;
    int s_cur = 0, s_end = 0;
    V3<F> acc = mk<F>(0, 0, 0);
    Path<F> path;
    path.o = path.d = path.atten = mk<F>(0, 0, 0);
    path.tm = 0;
    path.depth = 0;
    Rng rng = {0, 0, 0};

    for (;;) {
        // ---------------- unit hand-out --------------------------------------------------------------
        bool need = !have && !retired;
        uint64_t want = __ballot(need);
        while (want != 0ull) {
            if (pool_next == pool_end) {
                if (dry) break;
                // guided: half of an even share of what is left, between one and eight units per group
                const uint32_t left = n_units > last_base ? n_units - last_base : 0u;
                uint32_t batch = left / (2u * n_waves);
                batch = batch < kGroups ? kGroups : (batch > 8u * kGroups ? 8u * kGroups : batch);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.tail_count + 1, batch);
                base = __builtin_amdgcn_readfirstlane(base);
                last_base = base;
                if (base >= n_units) {
                    dry = true;
                    break;
                }
                pool_next = base;
                pool_end = n_units - base < batch ? n_units : base + batch;
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t rank = (uint32_t)__popcll(want & lanes_below_group) / (uint32_t)G; // needy groups before mine
            if (need && rank < avail) {
                load_unit<F>(P, P.tail_units[pool_next + rank], task, s_cur, s_end, need_ray, single, out_index, acc, path, rng);
                have = true;
                need = false;
            }
            const uint32_t needy = (uint32_t)__popcll(want) / (uint32_t)G;
            pool_next += needy < avail ? needy : avail;
            want = __ballot(need);
        }
        if (need) retired = true;
        if (__ballot(have) == 0ull) break;

        if (have) {
            if (need_ray) {
                need_ray = false;
                int px_i, px_j, sf, se;
                task_decode<F>(P, task, px_i, px_j, sf, se);
                camera_ray<F>(P, px_i, px_j, s_cur, rng, path);
            }
            bool done = false;
            V3<F> radiance = mk<F>(0, 0, 0);
            if (P.max_depth <= 0)
                done = true;
            else {
                n_segments += 1;
                const F a = vlen2<F>(path.d);
                HitInfo<F> lb;
                lb.t = Limits<F>::inf();
                lb.idx = -1;
                // The split scan + butterfly equals the sequential scan whenever no root is NaN; with a
                // finite scene (checked on the host) that holds for every ray whose magnitudes cannot
                // overflow the discriminant.  Anything else (non-finite or absurd rays) takes the plain
                // sequential scan on every lane: the reference's semantics, NaN behaviour included.
                const F o2 = path.o.x * path.o.x + path.o.y * path.o.y + path.o.z * path.o.z;
                const bool sane = a >= Limits<F>::coop_tiny() && a <= Limits<F>::coop_big() && o2 <= Limits<F>::coop_big() && ffabs(path.tm) <= Limits<F>::coop_big();
                if (sane) {
                    // FILTER: the conservative 7-FMA test of the render kernel's phase 1 first, the exact
                    // test (record from HBM) only for its candidates; otherwise the exact test throughout.
                    // (padding records: thr = +inf / r*r = -inf, never a hit)
                    FilterRay fr = {0, 0, 0, 0, 0, 0, Limits<float>::inf()};
                    if (FILTER) fr = make_filter_ray(path, a);
                    constexpr int U = 4; // loads in flight per lane
                    for (int p0 = sub; p0 < n_sph_pad; p0 += U * G) {
                        SphereHot<ST> g[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) g[u] = sph_tab[p0 + u * G < n_sph_pad ? p0 + u * G : 0];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int idx = p0 + u * G;
                            if (idx >= n_sph_pad) continue;
                            if (FILTER) {
                                if (!(filter_value(fr, (float)g[u].cx, (float)g[u].cy, (float)g[u].cz) < (float)g[u].r2)) { // the r2 slot holds thr
                                    const SphereHot<F> h = exact_tab[idx];
                                    refine_sphere<F>(h.cx, h.cy, h.cz, h.r2, path, a, t_min, idx, lb);
                                }
                            }
                            else
                                refine_sphere<F>((F)g[u].cx, (F)g[u].cy, (F)g[u].cz, (F)g[u].r2, path, a, t_min, idx, lb);
                        }
                    }
                    for (int q = sub; q < n_msph; q += G) {
                        const MovingSphereRec<F> ms = P.msph[q];
                        const V3<F> cen = msphere_center<F>(ms, path.tm);
                        refine_sphere<F>(cen.x, cen.y, cen.z, ms.r2, path, a, t_min, msph_base + q, lb);
                    }
                    for (int q = sub; q < n_tri; q += G) {
                        F tt;
                        if (triangle_test<F, true>(P.tri[q], path, t_min, lb.t, tt)) {
                            lb.t = tt;
                            lb.idx = tri_base + q;
                        }
                    }
                    // rank: sphere-like -> its index (later wins ties); triangle -> negative (earlier wins, loses to spheres)
                    int rank = lb.idx < tri_base ? lb.idx : -lb.idx - 1;
#pragma unroll
                    for (int off = G / 2; off >= 1; off >>= 1) { // stays inside the group
                        const F ot = __shfl_xor(lb.t, off);
                        const int oi = __shfl_xor(lb.idx, off);
                        const int orank = __shfl_xor(rank, off);
                        if (ot < lb.t || (ot == lb.t && orank > rank)) {
                            lb.t = ot;
                            lb.idx = oi;
                            rank = orank;
                        }
                    }
                }
                else {
                    lb = sequential_closest_hit<F>(P, path, a, t_min); // hittable_list.h:95-117, as it stands
                }
                done = shade<F>(P, lb, path, rng, radiance);
            }
            if (done) {
                if (P.per_sample) { // (every sample to its own slot: finalize_kernel adds them up)
                    if (sub == 0) {
                        F *o = P.out + ((size_t)task_pixel<F>(P, task) * (size_t)P.spp + (size_t)s_cur) * 3;
                        o[0] = radiance.x, o[1] = radiance.y, o[2] = radiance.z;
                    }
                }
                else
                    acc = single ? radiance : vadd<F>(acc, radiance); // units k >= 1 deliver the sample itself
                s_cur += 1;
                need_ray = true;
                if (s_cur >= s_end) {
                    if (sub == 0 && !P.per_sample) {
                        F *o = P.tail_rad + (size_t)out_index * 3;
                        o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
                    }
                    have = false;
                }
            }
        }
    }
    if (P.collect_stats) {
        // one count per group, summed over the wave's groups by lane 0
        uint32_t wave_segments = sub == 0 ? n_segments : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wave_segments += __shfl_xor(wave_segments, off);
        if (lane == 0 && wave_segments) {
            atomicAdd(&P.counters[0], (unsigned long long)wave_segments);
            atomicAdd(&P.counters[3], (unsigned long long)wave_segments); // the tail kernel tests every primitive
        }
    }
#ifdef RRTX_DIAG
    {
        uint32_t wave_segments = sub == 0 ? n_segments : 0u;
        for (int off = 32; off >= 1; off >>= 1) wave_segments += __shfl_xor(wave_segments, off);
        if (lane == 0) {
            const uint32_t wid = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
            unsigned long long *d = P.diag + (size_t)(65536 + wid) * 8;
            d[0] = diag_t0, d[2] = __builtin_amdgcn_s_memrealtime(), d[3] = wave_segments, d[4] = P.tail_count[0];
        }
    }
#endif
}

// Adds the tail kernel's results of each parked item in sample order (the order rrt.cu:115 adds them)
// and stores the task's partial sum.  One thread per item.
template <typename F> __global__ void __launch_bounds__(256) tail_sum_kernel(const KernelParams<F> P)
{
    if (P.per_sample) return; // (the units stored their samples themselves: there are no per-task sums)
    const uint32_t n_items = *P.tail_count;
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < n_items; item += gridDim.x * blockDim.x) {
        const TailItem<F> it = P.tail_items[item];
        int px_i, px_j, s_first, s_end;
        task_decode<F>(P, it.task, px_i, px_j, s_first, s_end);
        const int remaining = s_end - it.s_cur;
        const int units = remaining < kTailSplit ? remaining : kTailSplit;
        const F *r = P.tail_rad + (size_t)item * kTailSplit * 3;
        F ax = r[0], ay = r[1], az = r[2]; // unit 0 continued the item's partial sum
        for (int k = 1; k < units; ++k) {
            ax = ax + r[3 * k + 0];
            ay = ay + r[3 * k + 1];
            az = az + r[3 * k + 2];
        }
        F *o = task_slot<F>(P, it.task);
        o[0] = ax, o[1] = ay, o[2] = az;
    }
}

// ---------------------------------------------------------------------------------------------
// Camera-ray candidate lists.  One thread per pixel of this shard: which static spheres can ANY camera
// ray of the pixel hit?  camera::get_ray (camera.h:31-38) shoots from a point L of the lens disk
// (|L - origin| <= lens_radius) through a point T of the focus plane,
// T = lower_left_corner + s horizontal + t vertical with (s, t) in the pixel's cell
// [i/(W-1), (i+1)/(W-1)] x [j/(H-1), (j+1)/(H-1)].  Against the central ray (L0 = origin, T0 = cell
// centre, D = T0 - L0) a point of such a ray at parameter x deviates by at most
// dev(x) = |1 - x| Rl + |x| rho, with Rl = lens_radius and rho the cell's half extent.  If the ray
// touches sphere (c, r), then with perp = distance(c, central line), x* = the parameter of the foot
// point and kappa = (Rl + rho) / |D| < 1:      perp sqrt(1 - kappa^2) <= r + dev(x*).
// The kernel lists every sphere passing that test with 0.1 % + 1e-5 (|c - L0| + r) of slack (the
// rays are formed in fp32/fp64, the bound in exact arithmetic); up to kPlistCap per pixel, in index
// order; more (or kappa >= 0.9, or a degenerate camera) marks the pixel 0xFFFF = always scan.
// Double arithmetic for both precisions: this runs once per scene.
// ---------------------------------------------------------------------------------------------
// The conservative test behind the lists: can a ray of the bundle around direction D (camera origin -> a point of
// the focus plane; the bundle's points there lie within `rho` of D's, its lens offsets within Rl) touch sphere
// (w = centre - camera origin, radius r)?  dd = |D|^2, inv_cos = 1 / sqrt(1 - ((Rl + rho) / |D|)^2).
// `eps` is the unit roundoff of the precision the rays are traced in: the reference's discriminant, evaluated in that
// precision, can be >= 0 for a line that MISSES the sphere - by up to sqrt(r^2 + m) - r with m = 32 eps (|oc|^2 + r^2)
// (the bound build_grid inflates its boxes by; rrtx_grid.h) - and a pixel's list must hold every sphere the sequential
// scan would report, not only those geometry says can be hit.  For a small sphere far away that term, which grows
// with the SQUARE of the distance, dwarfs any relative slack: r = 0.2 at 200 units in fp32 "hits" up to 0.4 away.
__device__ __forceinline__ bool bundle_may_hit(const double w[3], double r, const double D[3], double dd, double rho, double Rl, double inv_cos, double eps)
{
    const double w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double wd = w[0] * D[0] + w[1] * D[1] + w[2] * D[2];
    const double xs = wd / dd; // parameter of the foot point
    double perp2 = w2 - wd * wd / dd;
    if (perp2 < 0.0) perp2 = 0.0;
    const double dev = fabs(1.0 - xs) * Rl + fabs(xs) * rho;
    const double oc_max = sqrt(w2) + Rl; // |o - c| of any ray of the bundle
    const double r_eff = sqrt(r * r + 32.0 * eps * (oc_max * oc_max + r * r));
    const double reach = (r_eff + dev) * inv_cos * 1.001 + 1e-5 * (sqrt(w2) + fabs(r));
    return !(perp2 > reach * reach); // (NaN-safe: anything strange is listed)
}
// One block per strip of up to 256 pixels of a row.  Phase A: the block tests every sphere once against the
// bundle of the whole strip (the same test with the strip's footprint for rho) and compacts the survivors, in
// primitive order, into LDS; phase B: every pixel tests only those against its own bundle.  (The per-pixel loop
// over all spheres this replaces cost 1.2 ms per scene on final.txt and 88 ms for 40 000 spheres - more than the
// frame.)  A strip whose list does not fit falls back to that loop.
constexpr int kStripList = 4096;
template <typename F> __global__ void __launch_bounds__(256) primary_lists_kernel(const KernelParams<F> P, uint16_t *plist)
{
    __shared__ uint16_t strip_list[kStripList];
    __shared__ uint32_t wave_count[4];
    __shared__ uint32_t strip_total;
    const int strips_per_row = (P.W + 255) / 256;
    const uint32_t lr = blockIdx.x / (uint32_t)strips_per_row;
    const int i0 = (int)(blockIdx.x % (uint32_t)strips_per_row) * 256;
    const int n_in = P.W - i0 < 256 ? P.W - i0 : 256;
    const uint32_t tile = lr / (uint32_t)P.tile_rows;
    const int j = (int)((tile * (uint32_t)P.shard_count + (uint32_t)P.shard_rank) * (uint32_t)P.tile_rows + (lr - tile * (uint32_t)P.tile_rows));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double len_h = 0, len_v = 0;
    for (int k = 0; k < 3; ++k) len_h += (double)P.cam.horizontal[k] * (double)P.cam.horizontal[k], len_v += (double)P.cam.vertical[k] * (double)P.cam.vertical[k];
    const double px_h = sqrt(len_h) / (double)(P.W - 1), px_v = sqrt(len_v) / (double)(P.H - 1); // a pixel's size on the focus plane
    const double Rl = fabs((double)P.cam.lens_radius);
    const double eps_f = sizeof(F) == 4 ? 0x1p-24 : 0x1p-53;
    const double t0 = ((double)j + 0.5) / (double)(P.H - 1);
    auto direction = [&](double ic, double D[3], double &dd) {
        const double s0 = (ic + 0.5) / (double)(P.W - 1);
        dd = 0;
        for (int k = 0; k < 3; ++k) {
            D[k] = (double)P.cam.llc[k] + s0 * (double)P.cam.horizontal[k] + t0 * (double)P.cam.vertical[k] - (double)P.cam.origin[k];
            dd += D[k] * D[k];
        }
    };
    // ---- phase A: the strip's bundle
    bool strip_ok;
    {
        double D[3], dd;
        direction((double)i0 + 0.5 * (double)(n_in - 1), D, dd);
        const double dist = sqrt(dd), rho = 0.5 * (double)n_in * px_h + 0.5 * px_v, kappa = (Rl + rho) / dist;
        strip_ok = dist > 0.0 && kappa < 0.9 && dd < 1e300; // (block-uniform; also catches NaN)
        if (threadIdx.x == 0) strip_total = 0;
        __syncthreads();
        if (strip_ok) {
            const double inv_cos = 1.0 / sqrt(1.0 - kappa * kappa);
            for (int base = 0; base < P.n_sph; base += 256) {
                const int k = base + (int)threadIdx.x;
                bool hit = false;
                if (k < P.n_sph) {
                    const SphereHot<F> g = P.sph_hot[k];
                    const double w[3] = {(double)g.cx - (double)P.cam.origin[0], (double)g.cy - (double)P.cam.origin[1], (double)g.cz - (double)P.cam.origin[2]};
                    hit = bundle_may_hit(w, (double)P.sph_cold[k].radius, D, dd, rho, Rl, inv_cos, eps_f);
                }
                const uint64_t m = __ballot(hit);
                if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t offset = strip_total;
                for (int q = 0; q < wave; ++q) offset += wave_count[q];
                const uint32_t total = strip_total + wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
                if (hit) {
                    const uint32_t slot = offset + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (slot < (uint32_t)kStripList) strip_list[slot] = (uint16_t)k;
                }
                __syncthreads();
                if (threadIdx.x == 0) strip_total = total;
                __syncthreads();
            }
        }
    }
    const uint32_t n_list = strip_total;
    const bool use_list = strip_ok && n_list <= (uint32_t)kStripList;
    // ---- phase B: every pixel's own bundle
    if ((int)threadIdx.x >= n_in) return;
    const int i = i0 + (int)threadIdx.x;
    uint16_t *out = plist + ((size_t)lr * (size_t)P.W + (size_t)i) * kPlistStride;
    double D[3], dd;
    direction((double)i, D, dd);
    const double dist = sqrt(dd), rho = 0.5 * px_h + 0.5 * px_v, kappa = (Rl + rho) / dist;
    if (!(dist > 0.0) || !(kappa < 0.9) || !(dd < 1e300)) { // also catches NaN
        out[0] = 0xFFFFu;
        return;
    }
    const double inv_cos = 1.0 / sqrt(1.0 - kappa * kappa);
    uint32_t count = 0;
    bool overflow = false;
    const int n_loop = use_list ? (int)n_list : P.n_sph;
    for (int q = 0; q < n_loop; ++q) {
        const int k = use_list ? (int)strip_list[q] : q;
        const SphereHot<F> g = P.sph_hot[k];
        const double w[3] = {(double)g.cx - (double)P.cam.origin[0], (double)g.cy - (double)P.cam.origin[1], (double)g.cz - (double)P.cam.origin[2]};
        if (bundle_may_hit(w, (double)P.sph_cold[k].radius, D, dd, rho, Rl, inv_cos, eps_f)) {
            if (count < (uint32_t)kPlistCap)
                out[1 + count] = (uint16_t)k;
            else
                overflow = true;
            count += 1;
        }
    }
    out[0] = overflow ? (uint16_t)0xFFFFu : (uint16_t)count;
}

// ---------------------------------------------------------------------------------------------
// The sky split.  A pixel whose camera-ray candidate list is EMPTY cannot be hit by any camera ray of any of its samples (the lists are a proven superset,
// primary_lists_kernel): every sample of it is ONE segment that ends in the sky (rrt.cu:69-75).  Inside the render loop such a sample still costs a trip through the
// camera-ray section, a LIST pass and the shading of a miss - at a quarter of the lanes, between the scans and walks of the others.  Here the pixels of a scene are
// partitioned once (order_*_kernel: stable, non-empty lists first), the render kernel's queue holds the first kind only, and sky_tasks_kernel finishes the tasks of the
// second kind densely: one lane per task, nothing but camera rays (the generator and draws of the render kernel: camera_ray()) and the sky's colour, added up in sample
// order into the task's slot (or, in per-sample launches, stored sample by sample) - the sums the render kernel would have formed, bit for bit.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) order_count_kernel(const uint16_t *plist, uint32_t n_pixels, uint32_t *block_counts)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool first = q < n_pixels && plist[(size_t)q * kPlistStride] != 0u; // (0xFFFF - an overflowed list - is of the first kind)
    __shared__ uint32_t wave_n[4];
    const uint64_t m = __ballot(first);
    if ((threadIdx.x & 63u) == 0u) wave_n[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
}
__global__ void __launch_bounds__(1024) order_scan_kernel(uint32_t *block_counts, uint32_t n_blocks, uint32_t *total_first)
{
    // exclusive prefix sum of block_counts in place, by ONE block: each thread a contiguous run, the runs' sums scanned through LDS
    __shared__ uint32_t run_sum[1024];
    const uint32_t per = (n_blocks + 1023u) / 1024u, b0 = threadIdx.x * per < n_blocks ? threadIdx.x * per : n_blocks, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += block_counts[b];
    run_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < 1024; ++t) {
            const uint32_t v = run_sum[t];
            run_sum[t] = acc;
            acc += v;
        }
        *total_first = acc;
    }
    __syncthreads();
    uint32_t acc = run_sum[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t v = block_counts[b];
        block_counts[b] = acc;
        acc += v;
    }
}
__global__ void __launch_bounds__(256) order_scatter_kernel(const uint16_t *plist, uint32_t n_pixels, const uint32_t *block_offsets, const uint32_t *total_first, uint32_t *order)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool in = q < n_pixels, first = in && plist[(size_t)q * kPlistStride] != 0u;
    __shared__ uint32_t wave_n[4];
    const uint64_t m = __ballot(first);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) wave_n[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0; // pixels of the first kind in this block's earlier waves ...
    for (uint32_t w = 0; w < wave; ++w) before += wave_n[w];
    const uint32_t first_before = block_offsets[blockIdx.x] + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); // ... and below this lane
    if (!in) return;
    // position of q: among its kind, in pixel order; the second kind starts where the first ends
    order[first ? first_before : *total_first + (q - first_before)] = q;
}
template <typename F> __global__ void __launch_bounds__(256) sky_tasks_kernel(const KernelParams<F> P, uint32_t first_position, uint32_t n_positions)
{
    uint32_t n_samples = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_positions; i += gridDim.x * 256u) {
        const uint32_t task = queue_task(P, first_position + i);
        int pi, pj, s0, s1;
        task_decode<F>(P, task, pi, pj, s0, s1);
        V3<F> acc = mk<F>(0, 0, 0);
        for (int s = s0; s < s1; ++s) {
            Rng rng;
            Path<F> path;
            camera_ray<F>(P, pi, pj, s, rng, path);
            const V3<F> radiance = vmul<F>(path.atten, sky_from_t<F>(sky_t<F>(vunit<F>(path.d)))); // shade(), the branch of a miss: rrt.cu:69-75
            if (P.per_sample) {
                F *o = P.out + ((size_t)task_pixel<F>(P, task) * (size_t)P.spp + (size_t)s) * 3;
                o[0] = radiance.x, o[1] = radiance.y, o[2] = radiance.z;
            }
            else
                acc = vadd<F>(acc, radiance); // rrt.cu:115
        }
        if (!P.per_sample) {
            F *o = task_slot<F>(P, task);
            o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
        }
        n_samples += (uint32_t)(s1 - s0);
    }
    if (P.collect_stats) { // one segment per sample; ONE atomic per block (a per-wave atomic on one word caps a dense kernel at 88 waves / us: EXPERIMENTS.md, round 4)
        __shared__ uint32_t wave_n[4];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_samples += __shfl_xor(n_samples, off);
        if ((threadIdx.x & 63u) == 0u) wave_n[threadIdx.x >> 6] = n_samples;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&P.counters[0], (unsigned long long)wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// The first bounce, densely (KernelParams::first; scenes of spheres alone with candidate lists).  One lane per queued task; for each sample of the task the camera ray
// (camera_ray(): the generator and draws of the render loop), the reference's scan over the only spheres a camera ray of that pixel can meet - the pixel's candidate
// list in index order with the running closest hit: hittable_list.h:95-117 restricted to a proven superset, what a LIST pass of the render loop did -, and the bounce
// (shade(): rrt.cu:49-76).  What is left of the sample is ONE record (rrtx_device.h): the path ended, with its radiance, or it goes on - origin, direction, the material
// whose albedo is the attenuation, the position of the generator.  Neighbouring lanes hold neighbouring chunks of one pixel and walk the same list: all 64 lanes work,
// where the render loop served camera rays with a quarter of them between the scans and walks of the rest (round 4 measured the dense form of a camera ray at a third
// of its cost inside the loop: the sky split).  32 bytes per sample written, read once by the lane that owns the task: 15 GB for 1200x800 spp 500.
// ---------------------------------------------------------------------------------------------
template <typename F> __global__ void __launch_bounds__(256) first_bounce_kernel(const KernelParams<F> P, uint32_t n_positions)
{
    typedef F FV4 __attribute__((ext_vector_type(4)));
    typedef typename FirstCode<F>::type Code;
    __shared__ uint32_t list_lds[8][256]; // a thread's copy of its pixel's list (8 dwords), word k at [k][thread]: no bank conflicts, nobody else's business
    FV4 *const recs = (FV4 *)P.first;
    const F t_min = (F)0.001; // rrt.cpp:32
    uint32_t *const mine = &list_lds[0][threadIdx.x];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_positions; i += gridDim.x * 256u) {
        const uint32_t task = queue_task(P, i);
        int pi, pj, s0, s1;
        task_decode<F>(P, task, pi, pj, s0, s1);
        typedef uint32_t U4 __attribute__((ext_vector_type(4)));
        const U4 *pl4 = (const U4 *)(P.plist + (size_t)task_pixel<F>(P, task) * kPlistStride);
        const U4 lo4 = pl4[0], hi4 = pl4[1];
        mine[0 * 256] = lo4.x, mine[1 * 256] = lo4.y, mine[2 * 256] = lo4.z, mine[3 * 256] = lo4.w;
        mine[4 * 256] = hi4.x, mine[5 * 256] = hi4.y, mine[6 * 256] = hi4.z, mine[7 * 256] = hi4.w;
        const uint32_t count = lo4.x & 0xFFFFu; // halfword 0: the count, or 0xFFFF = "scan everything"
        for (int s = s0; s < s1; ++s) {
            FV4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
            uint32_t code = kFirstUnknown << 30;
            if (count != 0xFFFFu) {
                Rng rng;
                Path<F> path;
                camera_ray<F>(P, pi, pj, s, rng, path);
                const F a = vlen2<F>(path.d); // sphere.h:36
                HitInfo<F> best = {Limits<F>::inf(), -1};
                for (uint32_t k = 1; k <= count; ++k) {
                    const int idx = (int)((mine[(k >> 1) * 256] >> ((k & 1u) * 16u)) & 0xFFFFu);
                    const SphereHot<F> g = P.sph_hot[idx];
                    refine_sphere<F>(g.cx, g.cy, g.cz, g.r2, path, a, t_min, idx, best);
                }
                V3<F> radiance;
                const bool done = shade<F, true>(P, best, path, rng, radiance); // rrt.cu:49-76, the bounce at depth 0
                if (done)
                    p0.x = radiance.x, p0.y = radiance.y, p0.z = radiance.z, code = kFirstDone << 30;
                else if (rng.n <= 0x3FFFu) {
                    p0.x = path.o.x, p0.y = path.o.y, p0.z = path.o.z, p0.w = path.d.x, p1.x = path.d.y, p1.y = path.d.z;
                    code = (kFirstRay << 30) | (rng.n << 16) | (uint32_t)P.sph_cold[best.idx].mat;
                }
            }
            p1.z = __builtin_bit_cast(F, (Code)code);
            const uint32_t k = (uint32_t)(s - s0);
            recs[first_slot(task, k, (uint32_t)P.chunk, 0u)] = p0;
            recs[first_slot(task, k, (uint32_t)P.chunk, 1u)] = p1;
        }
    }
}

// Sums the per-task partials of each pixel in chunk order (fixed shape => same image for any
// number of devices).  Launched unless every pixel is a single task.
// kFinalizeGroup chunked pixels per block: their partial sums - [pixel][chunk][3], one contiguous slab - are staged through
// LDS with coalesced loads, then one thread per (pixel, channel) adds its chunks in order (the ORDER is part of the image: no
// tree reduction).  Read straight from HBM by those threads, 12 bytes here and 12 bytes a pixel further on, the same sums cost
// 0.51 instead of 0.20 ms at 1200x800 spp 500.  STAGED = false: the plain form, for slabs that do not fit.
constexpr int kFinalizeGroup = 16;
constexpr size_t kFinalizeLdsBytes = 48 * 1024;
template <typename F, bool STAGED> __global__ void __launch_bounds__(256) finalize_kernel(const F *__restrict__ partial, F *__restrict__ fb, FinalizeShape S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const uint32_t n_values = S.n_pixels * 3u, n_chunked = S.taper_pixel * 3u;
    uint32_t v_begin = 0; // values below this one were formed by the staged part
    if (STAGED) {
        F *const slab = (F *)dyn_lds;
        const uint32_t per_pixel = (uint32_t)S.chunks_per_pixel * 3u;
        const uint32_t group = (uint32_t)S.group;
        const uint32_t n_groups = (S.taper_pixel + group - 1) / group;
        for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
            const uint32_t q0 = g * group;
            const uint32_t n_px = S.taper_pixel - q0 < group ? S.taper_pixel - q0 : group;
            const F *src = partial + (size_t)q0 * per_pixel;
            for (uint32_t i = threadIdx.x; i < n_px * per_pixel; i += blockDim.x) slab[i] = src[i];
            __syncthreads();
            if (threadIdx.x < n_px * 3u) {
                const uint32_t q = threadIdx.x / 3u, ch = threadIdx.x - q * 3u;
                const F *p = slab + q * per_pixel + ch;
                F s = 0;
                for (int c = 0; c < S.chunks_per_pixel; ++c) s = s + p[c * 3];
                fb[(size_t)(q0 + q) * 3 + ch] = s;
            }
            __syncthreads();
        }
        v_begin = n_chunked;
    }
    // one thread per (pixel, channel) value
    const F *const samples = partial + (size_t)n_chunked * S.chunks_per_pixel;
    for (uint32_t v = v_begin + blockIdx.x * blockDim.x + threadIdx.x; v < n_values; v += gridDim.x * blockDim.x) {
        F s = 0;
        if (v < n_chunked) {
            const uint32_t q = v / 3u, ch = v - q * 3u;
            const F *p = partial + (size_t)q * (size_t)S.chunks_per_pixel * 3 + ch; // [pixel][chunk][3], see task_slot
            for (int c = 0; c < S.chunks_per_pixel; ++c) s = s + p[(size_t)c * 3];
        }
        else {
            // the sums a chunk task would have formed: 0 + s0 + s1 + ... per chunk, then chunk by chunk
            const uint32_t dq = (v - n_chunked) / 3u, ch = (v - n_chunked) - dq * 3u;
            const F *p = samples + ((size_t)dq * (size_t)S.spp) * 3 + ch;
            for (int first = 0; first < S.spp; first += S.chunk) {
                const int end = first + S.chunk < S.spp ? first + S.chunk : S.spp;
                F cs = 0;
                for (int k = first; k < end; ++k) cs = cs + p[(size_t)k * 3];
                s = s + cs;
            }
        }
        fb[v] = s;
    }
}

// Per-sample launches (KernelParams::per_sample): `samples` holds every sample's radiance, [pixel][sample][3]; a pixel's value is ONE running sum over its
// samples in sample order - rrt.cu:110-115: pixel_color(0, 0, 0), then += per sample.  A chain of spp dependent additions per (pixel, channel), 2.9 M chains for
// a 1200x800 frame: every thread of a block owns one chain (85 pixels x 3 channels to a block of 256) and keeps its running sum in a register while the block
// streams the pixels' samples through LDS in slices of T samples - a contiguous T x 12 bytes per pixel and slice, loaded with coalesced 4-byte accesses by all
// threads - and within a slice the values are read eight ahead of the eight additions that take them.  (The first version staged 8 whole pixels per block and ran
// 24 chains on 256 threads, each addition behind its own LDS read: 3.9 ms for the 5.8 GB of configuration 3; this form is bound by the read of that buffer.)
constexpr int kSampleChainPixels = 85;
template <typename F> __global__ void __launch_bounds__(256) finalize_samples_kernel(const F *__restrict__ samples, F *__restrict__ fb, uint32_t n_pixels, int spp, int slice)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    F *const slab = (F *)dyn_lds; // [pixel of the group][sample of the slice][3]
    for (uint32_t q0 = blockIdx.x * (uint32_t)kSampleChainPixels; q0 < n_pixels; q0 += gridDim.x * (uint32_t)kSampleChainPixels) {
        const uint32_t n_px = n_pixels - q0 < (uint32_t)kSampleChainPixels ? n_pixels - q0 : (uint32_t)kSampleChainPixels;
        const uint32_t q = threadIdx.x / 3u, ch = threadIdx.x - q * 3u; // this thread's chain (threads 255 and those of absent pixels only help loading)
        const bool mine = q < n_px;
        F s = 0;
        for (int t0 = 0; t0 < spp; t0 += slice) {
            const int nt = spp - t0 < slice ? spp - t0 : slice;
            const uint32_t per_px = (uint32_t)nt * 3u;
            for (uint32_t i = threadIdx.x; i < n_px * per_px; i += 256u) {
                const uint32_t pq = i / per_px, r = i - pq * per_px;
                slab[pq * (uint32_t)slice * 3u + r] = samples[((size_t)(q0 + pq) * (size_t)spp + (size_t)t0) * 3 + r];
            }
            __syncthreads();
            if (mine) {
                const F *p = slab + q * (uint32_t)slice * 3u + ch;
                int t = 0;
                for (; t + 8 <= nt; t += 8) {
                    F v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = p[(t + j) * 3];
#pragma unroll
                    for (int j = 0; j < 8; ++j) s = s + v[j];
                }
                for (; t < nt; ++t) s = s + p[t * 3];
            }
            __syncthreads();
        }
        if (mine) fb[(size_t)(q0 + q) * 3 + ch] = s;
    }
}

// Multi-device gather, last step (rrtx_group.cpp): `gathered` holds the compact row blocks of the N shards
// side by side (shard r from row GatherShape::row_off[r] on, its rows in ascending order); the frame's row j
// belongs to shard (j / T) mod N, where it is local row (j / (T N)) T + j mod T.  One thread per value; pure
// HBM traffic (read 12 B, write 12 B per pixel), 100 MB for a 4K frame.
template <typename F> __global__ void __launch_bounds__(256) deinterleave_kernel(const F *__restrict__ gathered, F *__restrict__ frame, GatherShape S)
{
    const uint64_t n = (uint64_t)S.row_values * (uint64_t)S.height;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(v / S.row_values), x = (uint32_t)(v - (uint64_t)j * S.row_values);
        frame[v] = gathered[(uint64_t)gather_source_row(S, j) * S.row_values + x];
    }
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from rrtx_api.cpp)
// ---------------------------------------------------------------------------------------------
template <typename F, bool FILTER, int LDSMODE, bool VERIFY, int ACCEL, bool RESUME = false, int SOV = 0> hipError_t launch_variant(const KernelParams<F> &P, int grid_blocks, size_t lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL((render_kernel<F, FILTER, LDSMODE, VERIFY, ACCEL, RESUME, SOV>), dim3(grid_blocks), dim3(LDSMODE == 3 ? mf_block_threads(sizeof(F)) : kBlockThreads), lds_bytes, stream, P);
    return hipGetLastError();
}
// kPlain's condition (render_kernel, SOV & 4)
template <typename F> bool launch_is_plain(const KernelParams<F> &P) { return P.taper_task_base >= P.total_tasks && P.pixel_order != nullptr && !P.per_sample; }
// the accelerated variants (render and resume passes): tables in LDS or in HBM, scenes of spheres alone or of every kind
template <typename F, bool FILTER, bool RESUME> hipError_t launch_accel(const KernelParams<F> &P, int grid_blocks, size_t alds, hipStream_t stream)
{
    const bool spheres_only = P.n_msph == 0 && P.n_tri == 0;
    if (!RESUME && FILTER && spheres_only && launch_is_plain(P)) { // (what a frame of a scene of spheres alone is by default: variants that know instead of asking)
        if (P.first != nullptr) return alds ? launch_variant<F, FILTER, 0, false, 2, false, 6>(P, grid_blocks, alds, stream) : launch_variant<F, FILTER, 0, false, 1, false, 6>(P, grid_blocks, 0, stream);
        return alds ? launch_variant<F, FILTER, 0, false, 2, false, 5>(P, grid_blocks, alds, stream) : launch_variant<F, FILTER, 0, false, 1, false, 5>(P, grid_blocks, 0, stream);
    }
    if (!RESUME && spheres_only && P.first != nullptr) // (the render pass of a launch with a first-bounce pre-pass; its resume pass, a few thousand paths, decides at run time)
        return alds ? launch_variant<F, FILTER, 0, false, 2, false, 2>(P, grid_blocks, alds, stream) : launch_variant<F, FILTER, 0, false, 1, false, 2>(P, grid_blocks, 0, stream);
    if (alds) return spheres_only ? launch_variant<F, FILTER, 0, false, 2, RESUME, 1>(P, grid_blocks, alds, stream) : launch_variant<F, FILTER, 0, false, 2, RESUME, 0>(P, grid_blocks, alds, stream);
    return spheres_only ? launch_variant<F, FILTER, 0, false, 1, RESUME, 1>(P, grid_blocks, 0, stream) : launch_variant<F, FILTER, 0, false, 1, RESUME, 0>(P, grid_blocks, 0, stream);
}
// bytes of LDS the accelerated variant wants for its tables (0: they stay in HBM)
template <typename F> size_t accel_lds_bytes(const KernelParams<F> &P)
{
    const size_t b = (size_t)P.n_sph_padded * sizeof(SphereHot<F>) + ((size_t)P.n_grid_cells + 1) * 4 + (((size_t)P.n_grid_prims * sizeof(GridPrim) + 15) & ~(size_t)15);
    return b <= (size_t)kLdsSceneBytes ? b : 0;
}
template <typename F> hipError_t launch_render(const KernelParams<F> &P, bool filter, int lds_mode, int grid_blocks, hipStream_t stream)
{
    const size_t lds = lds_mode ? (size_t)P.n_sph_padded * sizeof(SphereHot<float>) : 0; // (LDS modes 1, 2 exist for the filter only: fp32 records)
    if (P.grid_cell_start) { // accelerated closest hit; the scan (scalar loads) is its fallback
        const size_t alds = accel_lds_bytes<F>(P);
        if (P.verify_lists) return launch_variant<F, true, 0, true, 1>(P, grid_blocks, 0, stream);
        return filter ? launch_accel<F, true, false>(P, grid_blocks, alds, stream) : launch_accel<F, false, false>(P, grid_blocks, alds, stream);
    }
    if (P.verify_lists) return launch_variant<F, true, 0, true, 0>(P, grid_blocks, 0, stream); // test build: filter + scalar loads + list check
    if (!filter) return launch_variant<F, false, 0, false, 0>(P, grid_blocks, 0, stream); // the exact scan is the fallback: scalar loads only
    switch (lds_mode) {
    case 1: return launch_variant<F, true, 1, false, 0>(P, grid_blocks, lds, stream);
    case 2: return launch_variant<F, true, 2, false, 0>(P, grid_blocks, lds, stream);
    case 3: // the filter on the matrix cores: 64 bytes of f16 operands per sphere
        if (launch_is_plain(P)) return launch_variant<F, true, 3, false, 0, false, 4>(P, grid_blocks, (size_t)((P.n_sph_padded + 31) & ~31) * 64, stream);
        return launch_variant<F, true, 3, false, 0>(P, grid_blocks, (size_t)((P.n_sph_padded + 31) & ~31) * 64, stream);
    default: return launch_variant<F, true, 0, false, 0>(P, grid_blocks, 0, stream);
    }
}
template <typename F> hipError_t launch_primary_lists(const KernelParams<F> &P, uint16_t *plist, hipStream_t stream)
{
    const uint32_t blocks = (uint32_t)P.local_rows * (uint32_t)((P.W + 255) / 256); // one per strip of a row
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(primary_lists_kernel<F>, dim3(blocks), dim3(256), 0, stream, P, plist);
    return hipGetLastError();
}
// pixel_order[n_pixels] from plist, the count of the first kind behind the block offsets; scratch: ceil(n_pixels / 256) + 1 dwords
hipError_t launch_order_pixels(const uint16_t *plist, uint32_t n_pixels, uint32_t *scratch, uint32_t *order, hipStream_t stream)
{
    if (n_pixels == 0) return hipSuccess;
    const uint32_t blocks = (n_pixels + 255u) / 256u;
    hipLaunchKernelGGL(order_count_kernel, dim3(blocks), dim3(256), 0, stream, plist, n_pixels, scratch);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, stream, scratch, blocks, scratch + blocks);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, stream, plist, n_pixels, scratch, scratch + blocks, order);
    return hipGetLastError();
}
template <typename F> hipError_t launch_sky_tasks(const KernelParams<F> &P, uint32_t first_position, uint32_t n_positions, int num_cus, hipStream_t stream)
{
    if (n_positions == 0) return hipSuccess;
    uint32_t blocks = (n_positions + 255u) / 256u;
    const uint32_t cap = (uint32_t)num_cus * 16u; // (grid-stride: a few thousand blocks, one statistics atomic each)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sky_tasks_kernel<F>, dim3(blocks), dim3(256), 0, stream, P, first_position, n_positions);
    return hipGetLastError();
}
template <typename F> hipError_t launch_first_bounce(const KernelParams<F> &P, uint32_t n_positions, int num_cus, hipStream_t stream)
{
    if (n_positions == 0) return hipSuccess;
    uint32_t blocks = (n_positions + 255u) / 256u;
    const uint32_t cap = (uint32_t)num_cus * 16u; // (grid-stride: a few thousand blocks, one statistics atomic each)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(first_bounce_kernel<F>, dim3(blocks), dim3(256), 0, stream, P, n_positions);
    return hipGetLastError();
}
template hipError_t launch_first_bounce<float>(const KernelParams<float> &, uint32_t, int, hipStream_t);
template hipError_t launch_first_bounce<double>(const KernelParams<double> &, uint32_t, int, hipStream_t);
template hipError_t launch_sky_tasks<float>(const KernelParams<float> &, uint32_t, uint32_t, int, hipStream_t);
template hipError_t launch_sky_tasks<double>(const KernelParams<double> &, uint32_t, uint32_t, int, hipStream_t);
#ifndef RRTX_TAIL_GROUP
#define RRTX_TAIL_GROUP 8 // lanes per ray in the tail kernel (measured on final.txt spp 48: 32 -> 1.82, 16 -> 1.33, 8 -> 1.18, 4 -> 1.17 ms)
#endif
// the accelerated variants finish their parked work with a resume pass of themselves (lane per ray on
// the grid), then tail_sum_kernel; `grid_blocks` is the render grid
template <typename F> hipError_t launch_resume(const KernelParams<F> &P, bool filter, int grid_blocks, hipStream_t stream)
{
    const size_t alds = accel_lds_bytes<F>(P);
    hipError_t e = filter ? launch_accel<F, true, true>(P, grid_blocks, alds, stream) : launch_accel<F, false, true>(P, grid_blocks, alds, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tail_sum_kernel<F>, dim3(256), dim3(256), 0, stream, P);
    return hipGetLastError();
}
template <typename F> hipError_t launch_tail(const KernelParams<F> &P, bool filter, int grid_blocks, hipStream_t stream)
{
    const size_t lds = (size_t)P.n_sph_padded * (filter ? sizeof(SphereHot<float>) + sizeof(SphereHot<F>) : sizeof(SphereHot<F>));
    const bool in_lds = lds <= (size_t)kLdsSceneBytes;
    if (filter && in_lds)
        hipLaunchKernelGGL((tail_kernel<F, RRTX_TAIL_GROUP, true, true>), dim3(grid_blocks), dim3(kBlockThreads), lds, stream, P);
    else if (filter)
        hipLaunchKernelGGL((tail_kernel<F, RRTX_TAIL_GROUP, false, true>), dim3(grid_blocks), dim3(kBlockThreads), 0, stream, P);
    else if (in_lds)
        hipLaunchKernelGGL((tail_kernel<F, RRTX_TAIL_GROUP, true, false>), dim3(grid_blocks), dim3(kBlockThreads), lds, stream, P);
    else
        hipLaunchKernelGGL((tail_kernel<F, RRTX_TAIL_GROUP, false, false>), dim3(grid_blocks), dim3(kBlockThreads), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tail_sum_kernel<F>, dim3(256), dim3(256), 0, stream, P);
    return hipGetLastError();
}
template <typename F> hipError_t launch_finalize(const F *partial, F *fb, const FinalizeShape &S, hipStream_t stream)
{
    if (S.per_sample) { // one running sum per pixel over its samples, [pixel][sample][3]
        const int slice = (int)(kFinalizeLdsBytes / ((size_t)kSampleChainPixels * 3 * sizeof(F))); // 48 samples in fp32, 24 in fp64
        uint32_t blocks = (S.n_pixels + kSampleChainPixels - 1) / kSampleChainPixels;
        if (blocks > 16384u) blocks = 16384u;
        if (blocks < 1u) blocks = 1u;
        hipLaunchKernelGGL(finalize_samples_kernel<F>, dim3(blocks), dim3(256), (size_t)kSampleChainPixels * slice * 3 * sizeof(F), stream, partial, fb, S.n_pixels, S.spp, slice);
        return hipGetLastError();
    }
    // (as many pixels per block as fit the LDS budget, 16 at the most: 16 for 63 chunk sums a pixel, 8 / 4 for 500 samples a pixel in fp32 / fp64)
    const size_t per_pixel = (size_t)S.chunks_per_pixel * 3 * sizeof(F);
    int group = per_pixel ? (int)(kFinalizeLdsBytes / per_pixel) : 0;
    if (group > kFinalizeGroup) group = kFinalizeGroup;
    if (S.taper_pixel > 0 && S.chunks_per_pixel > 1 && group >= 1) {
        FinalizeShape G = S;
        G.group = group;
        uint32_t blocks = (S.taper_pixel + (uint32_t)group - 1) / (uint32_t)group;
        if (blocks > 16384u) blocks = 16384u;
        hipLaunchKernelGGL((finalize_kernel<F, true>), dim3(blocks), dim3(256), (size_t)group * per_pixel, stream, partial, fb, G);
        return hipGetLastError();
    }
    int blocks = (int)((S.n_pixels * 3u + 255u) / 256u);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((finalize_kernel<F, false>), dim3(blocks), dim3(256), 0, stream, partial, fb, S);
    return hipGetLastError();
}
template <typename F> hipError_t render_occupancy(const KernelParams<F> &P, bool filter, int lds_mode, int *blocks_per_cu)
{
    const size_t lds = lds_mode ? (size_t)P.n_sph_padded * sizeof(SphereHot<float>) : 0; // (LDS modes 1, 2 exist for the filter only: fp32 records)
    if (P.grid_cell_start) {
        const size_t alds = accel_lds_bytes<F>(P);
        // (the variants for scenes of spheres alone and the densely pairing ones are compiled under different launch bounds)
        const bool so = P.n_msph == 0 && P.n_tri == 0;
        auto occ = [&](auto kernel, size_t lds) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, kBlockThreads, lds); };
        if (!filter) {
            if (so) return alds ? occ(render_kernel<F, false, 0, false, 2, false, true>, alds) : occ(render_kernel<F, false, 0, false, 1, false, true>, 0);
            return alds ? occ(render_kernel<F, false, 0, false, 2, false, false>, alds) : occ(render_kernel<F, false, 0, false, 1, false, false>, 0);
        }
        if (so) return alds ? occ(render_kernel<F, true, 0, false, 2, false, true>, alds) : occ(render_kernel<F, true, 0, false, 1, false, true>, 0);
        return alds ? occ(render_kernel<F, true, 0, false, 2, false, false>, alds) : occ(render_kernel<F, true, 0, false, 1, false, false>, 0);
    }
    if (!filter) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, render_kernel<F, false, 0, false, 0>, kBlockThreads, 0);
    switch (lds_mode) {
    case 1: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, render_kernel<F, true, 1, false, 0>, kBlockThreads, lds);
    case 2: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, render_kernel<F, true, 2, false, 0>, kBlockThreads, lds);
    case 3: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, render_kernel<F, true, 3, false, 0>, mf_block_threads(sizeof(F)), (size_t)((P.n_sph_padded + 31) & ~31) * 64);
    default: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, render_kernel<F, true, 0, false, 0>, kBlockThreads, 0);
    }
}

template <typename F> hipError_t launch_deinterleave(const F *gathered, F *frame, const GatherShape &S, hipStream_t stream)
{
    const uint64_t n = (uint64_t)S.row_values * S.height;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384; // (grid-stride: 64 blocks per CU)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(deinterleave_kernel<F>, dim3((uint32_t)blocks), dim3(256), 0, stream, gathered, frame, S);
    return hipGetLastError();
}

template hipError_t launch_deinterleave<float>(const float *, float *, const GatherShape &, hipStream_t);
template hipError_t launch_deinterleave<double>(const double *, double *, const GatherShape &, hipStream_t);
template hipError_t launch_render<float>(const KernelParams<float> &, bool, int, int, hipStream_t);
template hipError_t launch_render<double>(const KernelParams<double> &, bool, int, int, hipStream_t);
template hipError_t launch_primary_lists<float>(const KernelParams<float> &, uint16_t *, hipStream_t);
template hipError_t launch_primary_lists<double>(const KernelParams<double> &, uint16_t *, hipStream_t);
template hipError_t launch_tail<float>(const KernelParams<float> &, bool, int, hipStream_t);
template hipError_t launch_resume<float>(const KernelParams<float> &, bool, int, hipStream_t);
template hipError_t launch_resume<double>(const KernelParams<double> &, bool, int, hipStream_t);
template hipError_t launch_tail<double>(const KernelParams<double> &, bool, int, hipStream_t);
template hipError_t launch_finalize<float>(const float *, float *, const FinalizeShape &, hipStream_t);
template hipError_t launch_finalize<double>(const double *, double *, const FinalizeShape &, hipStream_t);
template hipError_t render_occupancy<float>(const KernelParams<float> &, bool, int, int *);
template hipError_t render_occupancy<double>(const KernelParams<double> &, bool, int, int *);

} // namespace rrtx
