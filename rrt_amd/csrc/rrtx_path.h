// The arithmetic of the path — vectors, RNG, the exact primitive tests, task decoding, camera ray,
// shading, the conservative scan filter and the accelerated closest hit — as header-only templates.
// rrtx_kernels.hip compiles them for gfx950; tests/path_host_check.cpp compiles the SAME source for the
// host (g++ -ffp-contract=off) and checks the grid walk against the sequential scan there.  Every
// function follows the reference's operation order (file:line beside it): that order is part of the image.
#ifndef RRTX_PATH_H
#define RRTX_PATH_H

#include <stdint.h>

#include "rrtx_device.h"

#if defined(__HIPCC__)
#define RRTX_DEV __device__ __forceinline__
#define RRTX_CONST_AS __attribute__((address_space(4)))
#else
#define RRTX_DEV inline
#endif

namespace rrtx {

// ---------------------------------------------------------------------------------------------
// small vector helpers — operation order mirrors vec3.h (it is part of the fp32 image)
// ---------------------------------------------------------------------------------------------
template <typename F> struct V3 {
    F x, y, z;
};
template <typename F> RRTX_DEV V3<F> mk(F x, F y, F z) { return V3<F>{x, y, z}; }
template <typename F> RRTX_DEV V3<F> ld3(const F *p) { return V3<F>{p[0], p[1], p[2]}; }
#ifdef RRTX_CONST_AS
template <typename F> RRTX_DEV V3<F> ld3(const RRTX_CONST_AS F *p) { return V3<F>{p[0], p[1], p[2]}; }
#endif
template <typename F> RRTX_DEV V3<F> vadd(V3<F> a, V3<F> b) { return mk<F>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename F> RRTX_DEV V3<F> vsub(V3<F> a, V3<F> b) { return mk<F>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename F> RRTX_DEV V3<F> vmul(V3<F> a, V3<F> b) { return mk<F>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename F> RRTX_DEV V3<F> vscale(F t, V3<F> v) { return mk<F>(t * v.x, t * v.y, t * v.z); } // vec3.h:109
template <typename F> RRTX_DEV V3<F> vneg(V3<F> a) { return mk<F>(-a.x, -a.y, -a.z); }
template <typename F> RRTX_DEV V3<F> vdiv(V3<F> v, F t) { return vscale<F>((F)1 / t, v); }             // vec3.h:113
template <typename F> RRTX_DEV F vdot(V3<F> a, V3<F> b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // vec3.h:115
template <typename F> RRTX_DEV V3<F> vcross(V3<F> u, V3<F> v)                                         // vec3.h:117-121
{
    return mk<F>(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
template <typename F> RRTX_DEV F vlen2(V3<F> a) { return a.x * a.x + a.y * a.y + a.z * a.z; } // vec3.h:60

RRTX_DEV float fsqrt(float x) { return __builtin_sqrtf(x); } // correctly rounded (hipcc default)
RRTX_DEV double fsqrt(double x) { return __builtin_sqrt(x); }
// Approximate (about 1 ulp) reciprocal, square root and reciprocal square root: ONE instruction each on
// the device against ~10 for the IEEE forms.  For the geometry of the grid walk only (which cell comes
// next, when to stop), where errors of a few ulps are covered a thousand times over by the inflation of the
// cells' boxes and the walk's slack — never for path arithmetic, which must round like the reference's.
#if defined(__HIP_DEVICE_COMPILE__)
RRTX_DEV float approx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RRTX_DEV double approx_rcp(double x) { return __builtin_amdgcn_rcp(x); }
RRTX_DEV float approx_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RRTX_DEV double approx_sqrt(double x) { return __builtin_amdgcn_sqrt(x); }
RRTX_DEV float approx_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
RRTX_DEV double approx_rsqrt(double x) { return __builtin_amdgcn_rsq(x); }
#else
template <typename F> RRTX_DEV F approx_rcp(F x) { return (F)1 / x; }
template <typename F> RRTX_DEV F approx_sqrt(F x) { return fsqrt(x); }
template <typename F> RRTX_DEV F approx_rsqrt(F x) { return (F)1 / fsqrt(x); }
#endif
RRTX_DEV float ffabs(float x) { return __builtin_fabsf(x); }
RRTX_DEV double ffabs(double x) { return __builtin_fabs(x); }
RRTX_DEV float ffmin(float a, float b) { return __builtin_fminf(a, b); }
RRTX_DEV double ffmin(double a, double b) { return __builtin_fmin(a, b); }
template <typename F> RRTX_DEV V3<F> vunit(V3<F> v) { return vdiv<F>(v, fsqrt(vlen2(v))); } // vec3.h:125
// explicit fused multiply-add: used ONLY by the conservative scan filter (never on the exact path)
RRTX_DEV float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RRTX_DEV double ffma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// POW(1-cosine, 5) of material.h:108.  rrtc calls powf(); x^5 formed in double from a float x is
// exact up to 2 ulp(double) and rounds to the same float except within 2^-28 of a rounding
// boundary; the value only feeds the comparison against a uniform draw (material.h:89).
RRTX_DEV float pow5(float x)
{
    double d = (double)x;
    double d2 = d * d;
    return (float)(d2 * d2 * d);
}
RRTX_DEV double pow5(double x)
{
    double x2 = x * x;
    return x2 * x2 * x;
}

template <int N> struct IntC {
    static constexpr int value = N;
};
template <typename F> struct SphereUnroll;
template <> struct SphereUnroll<float> {
    static constexpr int value = kSphereUnroll; // 8 x 16 B = 32 SGPRs per block
};
template <> struct SphereUnroll<double> {
    static constexpr int value = kSphereUnroll / 2; // 4 x 32 B = 32 SGPRs per block
};
template <typename F> struct Limits;
template <> struct Limits<float> {
    static RRTX_DEV float inf() { return __builtin_huge_valf(); }
    static RRTX_DEV float big() { return 1e30f; }
    static RRTX_DEV float tiny() { return 1e-30f; }
    static RRTX_DEV float coop_big() { return 1e15f; }   // squares and products of these stay finite
    static RRTX_DEV float coop_tiny() { return 1e-15f; }
};
template <> struct Limits<double> {
    static RRTX_DEV double inf() { return __builtin_huge_val(); }
    static RRTX_DEV double big() { return 1e280; }
    static RRTX_DEV double tiny() { return 1e-280; }
    static RRTX_DEV double coop_big() { return 1e120; }
    static RRTX_DEV double coop_tiny() { return 1e-120; }
};

// ---------------------------------------------------------------------------------------------
// RNG (DESIGN.md "RNG"; oracle/rrt_oracle.cpp holds the CPU statement of the same generator)
// ---------------------------------------------------------------------------------------------
struct Rng {
    uint32_t k0, k1, n;
};
RRTX_DEV uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x21F0AAADu;
    x ^= x >> 15;
    x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
RRTX_DEV void rng_open(Rng &r, uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t z = ((uint64_t)pixel << 32) | (uint64_t)sample;
    z += (uint64_t)seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    r.k0 = (uint32_t)z;
    r.k1 = (uint32_t)(z >> 32);
    r.n = 0;
}
template <typename F> RRTX_DEV F rng_uniform(Rng &r);
template <> RRTX_DEV float rng_uniform<float>(Rng &r)
{
    uint32_t hi = mix32(r.k0 + r.n * 0x9E3779B9u) + r.k1;
    r.n += 1;
    return (float)(hi >> 8) * 0x1p-24f;
}
template <> RRTX_DEV double rng_uniform<double>(Rng &r)
{
    uint32_t hi = mix32(r.k0 + r.n * 0x9E3779B9u) + r.k1;
    uint32_t lo = mix32(r.k1 + r.n * 0x85EBCA6Bu) + r.k0;
    r.n += 1;
    return (double)(((uint64_t)hi << 21) | (uint64_t)(lo >> 11)) * 0x1p-53;
}
// rtweekend.h:70-74
template <typename F> RRTX_DEV F rng_range(Rng &r, F lo, F hi) { return lo + (hi - lo) * rng_uniform<F>(r); }

// vec3.h:136-143, components drawn x, y, z
template <typename F> RRTX_DEV V3<F> in_unit_sphere(Rng &r)
{
    V3<F> p;
    do {
        p.x = rng_range<F>(r, (F)-1, (F)1);
        p.y = rng_range<F>(r, (F)-1, (F)1);
        p.z = rng_range<F>(r, (F)-1, (F)1);
    } while (vlen2(p) >= 1);
    return p;
}

// ---------------------------------------------------------------------------------------------
// per-lane path state
// ---------------------------------------------------------------------------------------------
template <typename F> struct Path {
    V3<F> o, d;   // current ray (ray.h)
    F tm;         // ray time
    V3<F> atten;  // running attenuation, rrt.cu:46,58
    int depth;    // bounce index i of rrt.cu:47
};

template <typename F> struct HitInfo {
    F t;
    int idx; // unified primitive index: spheres [0,n_sph), moving [n_sph_padded, +n_msph), triangles after
};

#ifndef RRTX_SKIP_BEHIND
#define RRTX_SKIP_BEHIND 1 // 0: experiments (A/B of the early exit for spheres behind the origin)
#endif
// Exact per-candidate test, reference order.  Spheres: sphere.h:33-49.
// AS_IT_STANDS: without the early exit for spheres behind the origin.  That exit is proven for FINITE operands only; the callers that
// promise the reference's scan "as it stands" - sequential_closest_hit(), the fallback for rays outside the proven range, and the
// VERIFY comparisons - take rays whose discriminant can be inf - inf = NaN, and the reference then ACCEPTS a NaN root (both
// comparisons of sphere.h:43-48 are false for it), which the exit would turn into "no hit".
template <typename F, bool AS_IT_STANDS = false> RRTX_DEV void refine_sphere(F cx, F cy, F cz, F r2, const Path<F> &p, F a, F t_min, int idx, HitInfo<F> &best)
{
    F ocx = p.o.x - cx, ocy = p.o.y - cy, ocz = p.o.z - cz;
    F half_b = ocx * p.d.x + ocy * p.d.y + ocz * p.d.z;
    F c = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;
    F disc = half_b * half_b - a * c;
    if (disc < 0) return;
#if RRTX_SKIP_BEHIND
    if (!AS_IT_STANDS && half_b > 0 && c > 0) return; // behind the origin: both roots are below t_min, sphere.h:43-48 rejects them whatever they are (see sphere_unordered)
#endif
    F sq = fsqrt(disc);
    F root = (-half_b - sq) / a;
    if (root < t_min || best.t < root) {
        root = (-half_b + sq) / a;
        if (root < t_min || best.t < root) return;
    }
    best.t = root;
    best.idx = idx;
}

template <typename F, typename MR> RRTX_DEV V3<F> msphere_center(const MR &m, F tm) // moving_sphere.h:27-30; MR: MovingSphereRec<F> in any address space
{
    F s = (tm - m.t0) / m.dt;
    return mk<F>(m.c0[0] + s * m.dc[0], m.c0[1] + s * m.dc[1], m.c0[2] + s * m.dc[2]);
}

// triangle.h:35-75.  stage 0: up to the u/v rejections (phase 1); stage 1: full test (phase 2).
template <typename F, bool FULL, typename TR> RRTX_DEV bool triangle_test(const TR &tr, const Path<F> &p, F t_min, F t_max, F &t_out) // TR: TriangleRec<F> in any address space
{
    const F EPS = (F)0.0000001;
    V3<F> e1 = ld3<F>(tr.e1), e2 = ld3<F>(tr.e2);
    V3<F> h = vcross<F>(p.d, e2);
    F a = vdot<F>(e1, h);
    if (a > -EPS && a < EPS) return false;
    F f = (F)1.0 / a;
    V3<F> s = vsub<F>(p.o, ld3<F>(tr.v0));
    F u = vdot<F>(vscale<F>(f, s), h);
    if (u < (F)0.0 || u > (F)1.0) return false;
    V3<F> q = vcross<F>(s, e1);
    F v = vdot<F>(vscale<F>(f, p.d), q);
    if (v < (F)0.0 || u + v > (F)1.0) return false;
    if (!FULL) return true;
    F t = vdot<F>(vscale<F>(f, e2), q);
    if (t > EPS && (t > t_min) && (t < t_max)) {
        t_out = t;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// Pieces shared by the render kernel (one ray per lane) and the tail kernel (one ray per wave)
// ---------------------------------------------------------------------------------------------
// Work items.  Local pixels q < taper_pixel are cut into chunks_per_pixel tasks of `chunk` samples
// (task = q * chunks_per_pixel + c).  The LAST pixels of the queue (q >= taper_pixel) are cut into
// single-sample tasks (task = taper_task_base + (q - taper_pixel) * spp + s): when the queue runs dry a
// lane then holds at most one unfinished path instead of half a chunk of them, which is what the
// end of a launch used to wait for.  finalize_kernel adds those samples chunk by chunk in sample
// order, so the image does not depend on where the taper starts.
template <typename FD> RRTX_DEV uint32_t fdiv(uint32_t n, const FD &f) // n / f.d for n < 2^31
{
    const uint32_t q = (uint32_t)(((uint64_t)n * (uint64_t)f.m) >> 32) >> f.shift; // (v_mul_hi_u32 on the device)
    return f.is_one ? n : q;
}

// PLAIN (here and below): the launch is known to have no single-sample tasks at its end, a queue order (the sky split) and per-task sums - what a frame of a scene of
// spheres alone is by default.  The render kernel has variants compiled for that (a launch decides it once, the loop asked at every task: rrtx_kernels.hip, SOV).
template <typename F, bool PLAIN = false, typename PP> RRTX_DEV uint32_t task_pixel(const PP &P, uint32_t task)
{
    if (PLAIN) return fdiv(task, P.div_cpp);
    return task < P.taper_task_base ? fdiv(task, P.div_cpp) : P.taper_pixel + fdiv(task - P.taper_task_base, P.div_spp);
}

// task -> (pixel, first/last sample)
template <typename F, bool PLAIN = false, typename PP> RRTX_DEV void task_decode(const PP &P, uint32_t task, int &px_i, int &px_j, int &s_first, int &s_end)
{
    uint32_t q;
    if (PLAIN || task < P.taper_task_base) {
        q = fdiv(task, P.div_cpp);
        const uint32_t c = task - q * (uint32_t)P.chunks_per_pixel;
        s_first = (int)c * P.chunk;
        s_end = s_first + P.chunk < P.spp ? s_first + P.chunk : P.spp;
    }
    else {
        const uint32_t t = task - P.taper_task_base;
        const uint32_t dq = fdiv(t, P.div_spp);
        q = P.taper_pixel + dq;
        s_first = (int)(t - dq * (uint32_t)P.spp);
        s_end = s_first + 1;
    }
    const uint32_t lr = fdiv(q, P.div_w);
    px_i = (int)(q - lr * (uint32_t)P.W);
    const uint32_t tile = fdiv(lr, P.div_tile);
    px_j = (int)((tile * (uint32_t)P.shard_count + (uint32_t)P.shard_rank) * (uint32_t)P.tile_rows + (lr - tile * (uint32_t)P.tile_rows));
}

// Position in the work queue -> task (KernelParams::pixel_order: the sky split).  The chunks of a pixel stay together and in order; everything else - the slot a task's
// sum goes to, the pixel and samples it stands for - is told by the TASK, so the order is scheduling only: the image does not know it.
template <bool PLAIN = false, typename PP> RRTX_DEV uint32_t queue_task(const PP &P, uint32_t position)
{
    if (!PLAIN && P.pixel_order == nullptr) return position;
    const uint32_t slot = fdiv(position, P.div_cpp);
    return P.pixel_order[slot] * (uint32_t)P.chunks_per_pixel + (position - slot * (uint32_t)P.chunks_per_pixel);
}

// Where a task's partial sum goes: task-major, [task][3] - chunked pixels first ([pixel < taper_pixel][chunk][3]: with one
// chunk per pixel and no taper this IS the local frame), the single-sample tasks after them ([pixel - taper_pixel][sample][3]).
// The lanes of a wave are handed consecutive tasks, so the 12-byte stores of a wave - minutes of kernel time apart, but
// into the same few lines, which L2 holds until they are full - reach HBM as whole lines.  (Round 1 had the chunked part
// chunk-major, [chunk][pixel][3], for finalize_kernel's sake: every store then dirtied a line of its own, 2.54 GB written
// for 0.73 GB of sums - 3.5 x; finalize reads 756 contiguous bytes per pixel now and costs the same 0.2 ms.)
template <typename F, typename PP> RRTX_DEV F *task_slot(const PP &P, uint32_t task) { return P.out + (size_t)task * 3; }

// camera ray of sample `s` of pixel (i, j): rrt.cu:112-114, camera.h:31-38
template <typename F, typename PP> RRTX_DEV void camera_ray(const PP &P, int px_i, int px_j, int s, Rng &rng, Path<F> &path)
{
    rng_open(rng, P.seed, (uint32_t)(px_j * P.W + px_i), (uint32_t)s);
    const F u = ((F)px_i + rng_uniform<F>(rng)) / (F)(P.W - 1);
    const F v = ((F)px_j + rng_uniform<F>(rng)) / (F)(P.H - 1);
    F dx, dy;
    do { // random_in_unit_disk, vec3.h:127-134
        dx = rng_range<F>(rng, (F)-1, (F)1);
        dy = rng_range<F>(rng, (F)-1, (F)1);
    } while (dx * dx + dy * dy >= 1); // + 0*0 of the z component changes nothing
    const F rdx = P.cam.lens_radius * dx, rdy = P.cam.lens_radius * dy;
    const V3<F> offset = vadd<F>(vscale<F>(rdx, ld3<F>(P.cam.u)), vscale<F>(rdy, ld3<F>(P.cam.v)));
    const V3<F> org = ld3<F>(P.cam.origin);
    path.o = vadd<F>(org, offset);
    path.d = vsub<F>(vsub<F>(vadd<F>(vadd<F>(ld3<F>(P.cam.llc), vscale<F>(u, ld3<F>(P.cam.horizontal))), vscale<F>(v, ld3<F>(P.cam.vertical))), org), offset);
    path.tm = rng_range<F>(rng, P.cam.time0, P.cam.time1);
    path.atten = mk<F>(1, 1, 1);
    path.depth = 0;
}

// The sky's colour for t = 0.5 (unit(d).y + 1): rrt.cu:69-75 with the CPU build's scalar types (rrt.cpp:47-50).  One statement, shared by shade() and by the
// kernel that finishes the sky-only pixels (sky_tasks_kernel): same operations, same bits.
template <typename F> RRTX_DEV V3<F> sky_from_t(F t) { return vadd<F>(vscale<F>((F)1.0 - t, mk<F>((F)1.0, (F)1.0, (F)1.0)), vscale<F>(t, mk<F>((F)0.5, (F)0.7, (F)1.0))); }
template <typename F> RRTX_DEV F sky_t(const V3<F> &unit_dir) { return (F)0.5 * (unit_dir.y + (F)1.0); }

// One bounce given the closest hit (rrt.cu:49-76).  Returns true when the path ended, with its
// radiance; otherwise `path` is the scattered ray.
// SO ("spheres only"): the scene holds neither moving spheres nor triangles - the branches that tell the kinds apart are
// compiled out of the accelerated kernels (final.txt, use_bvh: 41.1 -> 38.5 ms; the list scan gains nothing: it handles the
// kinds in loops of their own).
template <typename F, bool SO = false> RRTX_DEV bool shade(const KernelParams<F> &P, const HitInfo<F> &best, Path<F> &path, Rng &rng, V3<F> &radiance)
{
    const int msph_base = P.n_sph_padded, tri_base = P.n_sph_padded + P.n_msph;
    radiance = mk<F>(0, 0, 0);
    // unit_vector(r.direction()): the sky (rrt.cu:71), metal (material.h:52) and dielectric (material.h:82)
    // each form it; once here, because a wave with all three kinds of lanes would pay the square root
    // and the division three times
    const V3<F> ud = vunit<F>(path.d);
    if (best.idx < 0) {
        // sky, rrt.cu:69-75
        radiance = vmul<F>(path.atten, sky_from_t<F>(sky_t<F>(ud)));
        return true;
    }
    // hit record: sphere.h:51-55 / moving_sphere.h:50-55 / triangle.h:64-67
    const V3<F> hp = vadd<F>(path.o, vscale<F>(best.t, path.d)); // ray.h:17
    V3<F> outward;
    int mat_idx;
    if (SO || best.idx < msph_base) {
        const SphereHot<F> g = P.sph_hot[best.idx];
        const SphereCold<F> cold = P.sph_cold[best.idx];
        outward = vdiv<F>(vsub<F>(hp, mk<F>(g.cx, g.cy, g.cz)), cold.radius);
        mat_idx = cold.mat;
    }
    else if (best.idx < tri_base) {
        const MovingSphereRec<F> m = P.msph[best.idx - msph_base];
        outward = vdiv<F>(vsub<F>(hp, msphere_center<F>(m, path.tm)), m.radius);
        mat_idx = m.mat;
    }
    else {
        const TriangleRec<F> &tr = P.tri[best.idx - tri_base];
        outward = ld3<F>(tr.n);
        mat_idx = tr.mat;
    }
    const bool front_face = vdot<F>(path.d, outward) < 0; // hittable.h:18
    const V3<F> n = front_face ? outward : vneg<F>(outward);
    const MaterialRec<F> m = P.mat[mat_idx];
    V3<F> new_d;
    bool scattered = true;
    V3<F> albedo = mk<F>(m.r, m.g, m.b);
    if (m.type != 2) {
        const V3<F> rs = in_unit_sphere<F>(rng); // both lambertian and metal draw it (material.h:24,54)
        if (m.type == 0) {
            // lambertian, material.h:21-32
            new_d = vadd<F>(n, vunit<F>(rs));
            const double tiny = 1e-8; // vec3.h:65 compares in double
            if (((double)ffabs(new_d.x) < tiny) && ((double)ffabs(new_d.y) < tiny) && ((double)ffabs(new_d.z) < tiny)) new_d = n;
        }
        else {
            // metal, material.h:50-57
            const V3<F> reflected = vsub<F>(ud, vscale<F>((F)2 * vdot<F>(ud, n), n)); // vec3.h:156
            new_d = vadd<F>(reflected, vscale<F>(m.param, rs));
            scattered = vdot<F>(new_d, n) > 0;
        }
    }
    else {
        // dielectric, material.h:76-96
        albedo = mk<F>((F)1.0, (F)1.0, (F)1.0);
        const F ratio = front_face ? ((F)1.0 / m.param) : m.param;
        const F cos_theta = ffmin(vdot<F>(vneg<F>(ud), n), (F)1.0);
        const F sin_theta = fsqrt((F)1.0 - cos_theta * cos_theta);
        bool reflect_it = ratio * sin_theta > (F)1.0;
        if (!reflect_it) { // the uniform is drawn only here (short-circuit ||, material.h:89)
            F r0 = ((F)1 - ratio) / ((F)1 + ratio);
            r0 = r0 * r0;
            const F refl = r0 + ((F)1 - r0) * pow5((F)1 - cos_theta);
            reflect_it = refl > rng_uniform<F>(rng);
        }
        if (reflect_it)
            new_d = vsub<F>(ud, vscale<F>((F)2 * vdot<F>(ud, n), n));
        else {
            // refract, vec3.h:158-164
            const F ct = ffmin(vdot<F>(vneg<F>(ud), n), (F)1.0);
            const V3<F> perp = vscale<F>(ratio, vadd<F>(ud, vscale<F>(ct, n)));
            const V3<F> par = vscale<F>(-fsqrt(ffabs((F)1.0 - vlen2<F>(perp))), n);
            new_d = vadd<F>(perp, par);
        }
    }
    if (!scattered) return true; // absorbed, rrt.cu:65
    path.atten = vmul<F>(path.atten, albedo); // rrt.cu:58
    path.o = hp;
    path.d = new_d; // time unchanged (material.h:29)
    path.depth += 1;
    return path.depth >= P.max_depth; // rrt.cu:47,78: radiance stays 0
}

// ---------------------------------------------------------------------------------------------
// Accelerated closest hit (the reference's counterpart is its BVH, bvh.h:167-175).
//
// The sequential scan's answer is order-independent for finite rays: primitive p offers the root
// t_p = (near >= t_min ? near : far) if that is >= t_min (sphere.h:43-48; triangle: its t, triangle.h:63),
// and the winner is the smallest t_p — at equal t the LAST sphere-like primitive (root == t_max is
// accepted), while a triangle never displaces an equal t (strict <), so the FIRST triangle wins and
// any sphere beats it.  consider() applies exactly that order to candidates arriving in any order,
// so it suffices to run the exact test on a superset of the primitives whose test can succeed: the
// "always" list, then the cells of a uniform grid the ray walks through front to back (3-D DDA),
// stopping `slack` beyond the closest hit so far.  The cells were filled with boxes inflated by far
// more than the exact test's error for rays that start within sqrt(far2) of the grid (DESIGN.md has
// the bound); rays with non-finite or absurd components — and the few distant rays that could still
// touch the grid — report false and take the list scan.  Returns true when `best` is final.
// ---------------------------------------------------------------------------------------------
template <typename F> RRTX_DEV void consider(F t, int idx, int tri_base, HitInfo<F> &best)
{
    const int r = idx < tri_base ? idx : -idx - 1, br = best.idx < tri_base ? best.idx : -best.idx - 1;
    if (t < best.t || (t == best.t && r > br)) {
        best.t = t;
        best.idx = idx;
    }
}
// A sphere whose discriminant is >= 0, waiting for its roots: the square root and the division (~50
// instructions) are kept out of the loops over primitives, where any one lane taking them costs the
// whole wave — a lane holds at most one such candidate and resolves it at the end of a cell (or
// when the next one turns up).
template <typename F> struct PendingRoot {
    int idx; // -1: none
    F half_b, disc;
};
template <typename F> RRTX_DEV void resolve_pending(PendingRoot<F> &pend, F a, F t_min, int tri_base, HitInfo<F> &best)
{
    if (pend.idx < 0) return;
    // sphere.h:41-49 without the dependence on the scan order (see above)
    const F sq = fsqrt(pend.disc);
    F root = (-pend.half_b - sq) / a;
    bool ok = true;
    if (root < t_min) {
        root = (-pend.half_b + sq) / a;
        ok = !(root < t_min);
    }
    if (ok) consider<F>(root, pend.idx, tri_base, best);
    pend.idx = -1;
}
template <typename F> RRTX_DEV void sphere_unordered(F cx, F cy, F cz, F r2, const Path<F> &p, F a, F t_min, int idx, int tri_base, HitInfo<F> &best, PendingRoot<F> &pend)
{
    // sphere.h:33-40
    const F ocx = p.o.x - cx, ocy = p.o.y - cy, ocz = p.o.z - cz;
    const F half_b = ocx * p.d.x + ocy * p.d.y + ocz * p.d.z;
    const F c = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;
    const F disc = half_b * half_b - a * c;
    if (disc < 0) return;
#if RRTX_SKIP_BEHIND
    // The sphere lies behind the origin (the ray starts outside it, c > 0, and moves away, half_b > 0): sqrt(disc) <= half_b - disc =
    // half_b^2 - a c with a c >= 0 in floating point as in the reals, and the square root is monotone and exact on a square - so the far root
    // (-half_b + sqrt) / a is <= 0 < t_min and the near one smaller still: the reference rejects both (sphere.h:43-48), whatever they are.
    if (half_b > 0 && c > 0) return;
#endif
    resolve_pending<F>(pend, a, t_min, tri_base, best); // (rare: two candidates in one cell)
    pend.idx = idx, pend.half_b = half_b, pend.disc = disc;
}
constexpr int kNoTriangles = 0x7fffffff; // consider()'s tri_base where no primitive is a triangle
template <typename F, bool SO = false, typename PP, typename HotTab>
RRTX_DEV void test_primitive(const PP &P, const HotTab &hot, int idx, const Path<F> &path, F a, F t_min, HitInfo<F> &best, PendingRoot<F> &pend)
{
    const int msph_base = P.n_sph_padded, tri_base = SO ? kNoTriangles : P.n_sph_padded + P.n_msph;
    if (SO || idx < msph_base) {
        const SphereHot<F> g = hot[idx];
        sphere_unordered<F>(g.cx, g.cy, g.cz, g.r2, path, a, t_min, idx, tri_base, best, pend);
    }
    else if (idx < tri_base) {
        const MovingSphereRec<F> ms = P.msph[idx - msph_base];
        const V3<F> cen = msphere_center<F>(ms, path.tm);
        sphere_unordered<F>(cen.x, cen.y, cen.z, ms.r2, path, a, t_min, idx, tri_base, best, pend);
    }
    else {
        F tt;
        if (triangle_test<F, true>(P.tri[idx - tri_base], path, t_min, Limits<F>::inf(), tt)) consider<F>(tt, idx, tri_base, best);
    }
}
// Walk state a lane carries from one iteration of the render loop to the next: the walk is done in
// slices of `max_cells` cells, because a wave otherwise waits for its longest walker — path lengths
// through the sphere layer are roughly exponential (mean 2.5 cells), and the longest of 64 such
// walks is ~12 cells: 31 % lane utilisation measured.  Lanes whose walk is over go on to shade and
// start their next segment while the long walkers continue.
// kWalkNeedsScan: a ray the order-independent rule is not proven for - the sequential scan decides.
// kWalkFarScan: a sane ray from beyond the grid's range that touches its fattened box - every primitive has to be
// tested exactly, in any order (the render kernel does that with the whole wave, see there).
enum { kWalkDone = 0, kWalkNeedsScan = 1, kWalkGoesOn = 2, kWalkFarScan = 3 };
template <typename F, bool SO = false, typename PP, typename HotTab, typename CellTab, typename PrimTab>
RRTX_DEV int accel_closest_hit(const PP &P, const HotTab &hot, const CellTab &cell_start, const PrimTab &cell_prims, const Path<F> &path, F a, F t_min, HitInfo<F> &best, bool resume,
                               uint32_t &walk_cell, F &walk_t_out, int max_cells)
{
    const F ox = path.o.x, oy = path.o.y, oz = path.o.z, dx = path.d.x, dy = path.d.y, dz = path.d.z;
    const F rx = ox - P.grid.center[0], ry = oy - P.grid.center[1], rz = oz - P.grid.center[2];
    const F dist2 = rx * rx + ry * ry + rz * rz;
    const F reach = P.grid.slack1 * (approx_sqrt(dist2) + P.grid.half_diag); // 1.5 sqrt(32 eps) (|o - centre| + half diagonal)
    const int tri_base_ = SO ? kNoTriangles : P.n_sph_padded + P.n_msph;
    PendingRoot<F> pend = {-1, 0, 0};
    const F o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    F inv[3], tmax[3];
    int ci[3];
    F t_out = walk_t_out;
    // (a component too small for 1 / d to be finite counts as parallel: (x - o) * inf would be NaN for x == o)
    bool par[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) par[k] = !(ffabs(d[k]) >= Limits<F>::coop_tiny()), inv[k] = par[k] ? Limits<F>::inf() : approx_rcp(d[k]);
    if (!resume) {
        {
            // the rays the unordered rule is proven for
            const F o2 = ox * ox + oy * oy + oz * oz;
            const bool ok = a >= Limits<F>::coop_tiny() && a <= P.grid.dir2_max && o2 <= Limits<F>::coop_big() && ffabs(path.tm) <= Limits<F>::coop_big() && dist2 <= Limits<F>::coop_big();
            if (!ok) return kWalkNeedsScan;
        }
#ifdef RRTX_CONST_AS
        const RRTX_CONST_AS uint32_t *always = (const RRTX_CONST_AS uint32_t *)P.grid_always; // constant address space: s_load, not a vector load of a uniform address
#else
        const uint32_t *always = P.grid_always;
#endif
        for (int i = 0; i < P.n_always; ++i) test_primitive<F, SO>(P, hot, (int)always[i], path, a, t_min, best, pend);
        resolve_pending<F>(pend, a, t_min, tri_base_, best);

        // Rays that start beyond `far`: their exact test can "hit" spheres the line misses by more than
        // the cells' inflation — but by less than sqrt(m), m = 32 eps (|o - c|^2 + r^2).  Almost
        // all of them (the bounce off the distant ground, up into the sky) miss the grid's box even when it
        // is blown up by that much: no gridded primitive can answer them.  The few that do not are scanned.
        const bool is_far = dist2 > P.grid.far2;
        const F fat = is_far ? reach + reach : (F)0;

        // clip the ray to the grid's box: [t_in, t_out]
        F t_in = 0;
        t_out = Limits<F>::inf();
        bool miss = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const F glo = P.grid.gmin[k] - fat, ghi = P.grid.gmax[k] + fat;
            if (!par[k]) {
                const F t1 = (glo - o[k]) * inv[k], t2 = (ghi - o[k]) * inv[k];
                const F lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
                t_in = lo > t_in ? lo : t_in;
                t_out = hi < t_out ? hi : t_out;
            }
            else if (o[k] < glo || o[k] > ghi)
                miss = true;
        }
        if (miss || !(t_in <= t_out * ((F)1 + (F)1e-3))) return kWalkDone; // (a hair of tolerance on the far side: the slab arithmetic rounds too)
        if (is_far) return kWalkFarScan;
        if (t_in > best.t + (P.grid.slack + reach) * approx_rsqrt(a)) return kWalkDone;
        // the cell of the entry point
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const F pk = o[k] + d[k] * t_in;
            int c = (int)((pk - P.grid.gmin[k]) * P.grid.inv_cell[k]);
            ci[k] = c < 0 ? 0 : (c > P.grid.dims[k] - 1 ? P.grid.dims[k] - 1 : c);
        }
    }
    else
        ci[0] = (int)(walk_cell & 1023u), ci[1] = (int)((walk_cell >> 10) & 1023u), ci[2] = (int)(walk_cell >> 20);
    const F slack_t = (P.grid.slack + reach) * approx_rsqrt(a);
    // the DDA's per-axis distances to the next cell boundary (from the cell, not accumulated: a resumed
    // walk must not depend on where it was interrupted)
#pragma unroll
    for (int k = 0; k < 3; ++k) tmax[k] = par[k] ? Limits<F>::inf() : (P.grid.gmin[k] + (F)(ci[k] + (d[k] > 0 ? 1 : 0)) * P.grid.cell[k] - o[k]) * inv[k];
    const F dtx = P.grid.cell[0] * ffabs(inv[0]), dty = P.grid.cell[1] * ffabs(inv[1]), dtz = P.grid.cell[2] * ffabs(inv[2]);
    const int sx = dx > 0 ? 1 : -1, sy = dy > 0 ? 1 : -1, sz = dz > 0 ? 1 : -1;
    // (a single loop whose trips either test a primitive or step a cell was tried: 59.9 vs 52.6 ms)
    for (int step = 0; step < max_cells; ++step) {
        const uint32_t cell = ((uint32_t)ci[2] * (uint32_t)P.grid.dims[1] + (uint32_t)ci[1]) * (uint32_t)P.grid.dims[0] + (uint32_t)ci[0];
        const uint32_t beg = cell_start[cell], end = cell_start[cell + 1];
        for (uint32_t k = beg; k < end; ++k) test_primitive<F, SO>(P, hot, (int)cell_prims[k], path, a, t_min, best, pend); // (fetching index k + 1 before testing entry k: 42.0 vs 41.1 ms)
        resolve_pending<F>(pend, a, t_min, tri_base_, best);
        // next cell: across the nearest boundary (branch-free: the three axes diverge otherwise)
        const bool ax = tmax[0] <= tmax[1] && tmax[0] <= tmax[2];
        const bool ay = !ax && tmax[1] <= tmax[2];
        const bool az = !ax && !ay;
        const F t_next = ax ? tmax[0] : (ay ? tmax[1] : tmax[2]);
        if (t_next > t_out || t_next > best.t + slack_t) return kWalkDone;
        ci[0] += ax ? sx : 0, ci[1] += ay ? sy : 0, ci[2] += az ? sz : 0;
        tmax[0] += ax ? dtx : (F)0, tmax[1] += ay ? dty : (F)0, tmax[2] += az ? dtz : (F)0;
        if ((uint32_t)ci[0] >= (uint32_t)P.grid.dims[0] || (uint32_t)ci[1] >= (uint32_t)P.grid.dims[1] || (uint32_t)ci[2] >= (uint32_t)P.grid.dims[2]) return kWalkDone;
    }
    walk_cell = (uint32_t)ci[0] | ((uint32_t)ci[1] << 10) | ((uint32_t)ci[2] << 20);
    walk_t_out = t_out;
    return kWalkGoesOn;
}

// ---------------------------------------------------------------------------------------------
// The same walk, cut differently: accel_walk_prepare() runs everything of a slice that does NOT depend on the
// slice's own tests - the always-list and the box clip of a new segment, then the DDA through up to `max_cells`
// cells - and returns the cells' entry lists as up to four (first, count) ranges of cell_prims instead of testing
// them; the caller tests those entries in ANY order with test_primitive() / consider() (the render kernel: all 64
// lanes of the wave share out the (ray, entry) pairs of all its lanes, see dense_candidates there) and then asks
// accel_walk_decide() whether the walk goes on.  Against accel_closest_hit() a lane may so test cells that lie
// beyond a hit found earlier in the same slice - a superset of its tests, hence, by the order-independence above,
// the same answer; where a slice ends the two make the same decision with the same `best`.
// A range holds up to kDenseCellMax entries (counts are packed in 8 bits); a fuller cell takes several ranges, one of more than
// four ranges' worth is tested here, by its lane alone.
// ---------------------------------------------------------------------------------------------
struct WalkRanges {
    uint32_t beg0, beg1, beg2, beg3; // first entry of each range (an index into cell_prims)
    uint32_t cnt;                    // entries per range, 8 bits each; ranges are filled from 0 up
    uint32_t steps;                  // cells the DDA went through for them (statistics)
};
constexpr uint32_t kDenseCellMax = 250; // (counts are packed in 8 bits per range)
constexpr int kDenseRanges = 4;
template <typename F, bool SO = false, typename PP, typename HotTab, typename CellTab, typename PrimTab>
RRTX_DEV int accel_walk_prepare(const PP &P, const HotTab &hot, const CellTab &cell_start, const PrimTab &cell_prims, const Path<F> &path, F a, F t_min, HitInfo<F> &best, bool resume,
                                uint32_t &walk_cell, F &walk_t_out, int max_cells, WalkRanges &R, F &t_last, F &slack_out, bool &ended, const uint8_t *coarse = nullptr)
{
    R.beg0 = R.beg1 = R.beg2 = R.beg3 = 0u, R.cnt = 0u, R.steps = 0u;
    t_last = 0, slack_out = 0, ended = true;
    const F ox = path.o.x, oy = path.o.y, oz = path.o.z, dx = path.d.x, dy = path.d.y, dz = path.d.z;
    const F rx = ox - P.grid.center[0], ry = oy - P.grid.center[1], rz = oz - P.grid.center[2];
    const F dist2 = rx * rx + ry * ry + rz * rz;
    const F reach = P.grid.slack1 * (approx_sqrt(dist2) + P.grid.half_diag);
    const int tri_base_ = SO ? kNoTriangles : P.n_sph_padded + P.n_msph;
    PendingRoot<F> pend = {-1, 0, 0};
    const F o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    F inv[3], tmax[3];
    int ci[3];
    F t_out = walk_t_out;
    bool par[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) par[k] = !(ffabs(d[k]) >= Limits<F>::coop_tiny()), inv[k] = par[k] ? Limits<F>::inf() : approx_rcp(d[k]);
    if (!resume) {
        {
            const F o2 = ox * ox + oy * oy + oz * oz;
            const bool ok = a >= Limits<F>::coop_tiny() && a <= P.grid.dir2_max && o2 <= Limits<F>::coop_big() && ffabs(path.tm) <= Limits<F>::coop_big() && dist2 <= Limits<F>::coop_big();
            if (!ok) return kWalkNeedsScan;
        }
#ifdef RRTX_CONST_AS
        const RRTX_CONST_AS uint32_t *always = (const RRTX_CONST_AS uint32_t *)P.grid_always;
#else
        const uint32_t *always = P.grid_always;
#endif
        for (int i = 0; i < P.n_always; ++i) test_primitive<F, SO>(P, hot, (int)always[i], path, a, t_min, best, pend);
        resolve_pending<F>(pend, a, t_min, tri_base_, best);
        const bool is_far = dist2 > P.grid.far2;
        const F fat = is_far ? reach + reach : (F)0;
        F t_in = 0;
        t_out = Limits<F>::inf();
        bool miss = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const F glo = P.grid.gmin[k] - fat, ghi = P.grid.gmax[k] + fat;
            if (!par[k]) {
                const F t1 = (glo - o[k]) * inv[k], t2 = (ghi - o[k]) * inv[k];
                const F lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
                t_in = lo > t_in ? lo : t_in;
                t_out = hi < t_out ? hi : t_out;
            }
            else if (o[k] < glo || o[k] > ghi)
                miss = true;
        }
        if (miss || !(t_in <= t_out * ((F)1 + (F)1e-3))) return kWalkDone;
        if (is_far) return kWalkFarScan;
        if (t_in > best.t + (P.grid.slack + reach) * approx_rsqrt(a)) return kWalkDone;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const F pk = o[k] + d[k] * t_in;
            int c = (int)((pk - P.grid.gmin[k]) * P.grid.inv_cell[k]);
            ci[k] = c < 0 ? 0 : (c > P.grid.dims[k] - 1 ? P.grid.dims[k] - 1 : c);
        }
    }
    else
        ci[0] = (int)(walk_cell & 1023u), ci[1] = (int)((walk_cell >> 10) & 1023u), ci[2] = (int)(walk_cell >> 20);
    const F slack_t = (P.grid.slack + reach) * approx_rsqrt(a);
    slack_out = slack_t;
#pragma unroll
    for (int k = 0; k < 3; ++k) tmax[k] = par[k] ? Limits<F>::inf() : (P.grid.gmin[k] + (F)(ci[k] + (d[k] > 0 ? 1 : 0)) * P.grid.cell[k] - o[k]) * inv[k];
    const F dtx = P.grid.cell[0] * ffabs(inv[0]), dty = P.grid.cell[1] * ffabs(inv[1]), dtz = P.grid.cell[2] * ffabs(inv[2]);
    const int sx = dx > 0 ? 1 : -1, sy = dy > 0 ? 1 : -1, sz = dz > 0 ? 1 : -1;
    // The cells come in batches of kWalkBatch: first the DDA alone runs through the batch (which cells, where each ends, whether the
    // walk ends there - nothing of that depends on what the cells hold), then the batch's list headers are fetched TOGETHER, then they
    // are taken up in order.  With the grid in HBM a header is a dependent load of several hundred clocks, and a walk through a large,
    // mostly empty grid (a mesh: slices of 16 cells) used to pay them one after the other.
#ifndef RRTX_WALK_BATCH
#define RRTX_WALK_BATCH 4
#endif
    constexpr int kWalkBatch = RRTX_WALK_BATCH;
    int nr = 0, steps_left = max_cells;
    uint32_t pos = (uint32_t)ci[0] | ((uint32_t)ci[1] << 10) | ((uint32_t)ci[2] << 20);
    ended = false;
    while (steps_left > 0 && nr < kDenseRanges && !ended) {
        uint32_t cellv[kWalkBatch], nextv[kWalkBatch];
        F tnv[kWalkBatch];
        bool endv[kWalkBatch];
        int m = 0;
        bool stop = false;
#pragma unroll
        for (int q = 0; q < kWalkBatch; ++q) {
            cellv[q] = 0u, nextv[q] = 0u, tnv[q] = 0, endv[q] = false;
            if (q < steps_left && !stop) {
                // (an empty block of 4 x 4 x 4 cells is crossed in one step: GridRec::coarse_off)
                bool jump = false;
                int cc[3] = {0, 0, 0};
                if (coarse != nullptr) {
                    cc[0] = ci[0] >> 2, cc[1] = ci[1] >> 2, cc[2] = ci[2] >> 2;
                    jump = coarse[((uint32_t)cc[2] * (uint32_t)P.grid.coarse_dims[1] + (uint32_t)cc[1]) * (uint32_t)P.grid.coarse_dims[0] + (uint32_t)cc[0]] == 0;
                }
                F t_next;
                bool e;
                if (jump) {
                    cellv[q] = 0xFFFFFFFFu; // no list to fetch
                    F tcb[3];
                    int bk[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        bk[k] = d[k] > 0 ? ((cc[k] + 1) << 2) : (cc[k] << 2); // the block's face the ray leaves through, as a cell boundary
                        tcb[k] = par[k] ? Limits<F>::inf() : (P.grid.gmin[k] + (F)bk[k] * P.grid.cell[k] - o[k]) * inv[k];
                    }
                    const int axis = tcb[0] <= tcb[1] && tcb[0] <= tcb[2] ? 0 : (tcb[1] <= tcb[2] ? 1 : 2);
                    t_next = axis == 0 ? tcb[0] : (axis == 1 ? tcb[1] : tcb[2]);
                    e = t_next > t_out || t_next > best.t + slack_t;
                    if (!e) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            if (k == axis)
                                ci[k] = d[k] > 0 ? bk[k] : bk[k] - 1;
                            else { // the other two stay inside the block they were in
                                const int c = (int)((o[k] + d[k] * t_next - P.grid.gmin[k]) * P.grid.inv_cell[k]);
                                const int lo_c = cc[k] << 2, hi_c = (cc[k] << 2) + 3 < P.grid.dims[k] - 1 ? (cc[k] << 2) + 3 : P.grid.dims[k] - 1;
                                ci[k] = c < lo_c ? lo_c : (c > hi_c ? hi_c : c);
                            }
                        }
                        e = (uint32_t)ci[0] >= (uint32_t)P.grid.dims[0] || (uint32_t)ci[1] >= (uint32_t)P.grid.dims[1] || (uint32_t)ci[2] >= (uint32_t)P.grid.dims[2];
#pragma unroll
                        for (int k = 0; k < 3; ++k) tmax[k] = par[k] ? Limits<F>::inf() : (P.grid.gmin[k] + (F)(ci[k] + (d[k] > 0 ? 1 : 0)) * P.grid.cell[k] - o[k]) * inv[k];
                    }
                }
                else {
                    cellv[q] = ((uint32_t)ci[2] * (uint32_t)P.grid.dims[1] + (uint32_t)ci[1]) * (uint32_t)P.grid.dims[0] + (uint32_t)ci[0];
                    const bool ax = tmax[0] <= tmax[1] && tmax[0] <= tmax[2];
                    const bool ay = !ax && tmax[1] <= tmax[2];
                    const bool az = !ax && !ay;
                    t_next = ax ? tmax[0] : (ay ? tmax[1] : tmax[2]);
                    e = t_next > t_out || t_next > best.t + slack_t;
                    if (!e) {
                        ci[0] += ax ? sx : 0, ci[1] += ay ? sy : 0, ci[2] += az ? sz : 0;
                        tmax[0] += ax ? dtx : (F)0, tmax[1] += ay ? dty : (F)0, tmax[2] += az ? dtz : (F)0;
                        e = (uint32_t)ci[0] >= (uint32_t)P.grid.dims[0] || (uint32_t)ci[1] >= (uint32_t)P.grid.dims[1] || (uint32_t)ci[2] >= (uint32_t)P.grid.dims[2];
                    }
                }
                tnv[q] = t_next, endv[q] = e, nextv[q] = (uint32_t)ci[0] | ((uint32_t)ci[1] << 10) | ((uint32_t)ci[2] << 20);
                m = q + 1, stop = e;
            }
        }
        uint32_t hb[kWalkBatch], he[kWalkBatch];
#pragma unroll
        for (int q = 0; q < kWalkBatch; ++q) hb[q] = q < m && cellv[q] != 0xFFFFFFFFu ? cell_start[cellv[q]] : 0u, he[q] = q < m && cellv[q] != 0xFFFFFFFFu ? cell_start[cellv[q] + 1u] : 0u;
#pragma unroll
        for (int q = 0; q < kWalkBatch; ++q) {
            if (q < m && nr < kDenseRanges) {
                const uint32_t beg = hb[q], n = he[q] - hb[q];
                if (n > kDenseCellMax * (uint32_t)kDenseRanges) { // more entries than four ranges hold: the cell's lane tests it alone, here and now
                    for (uint32_t k = beg; k < he[q]; ++k) test_primitive<F, SO>(P, hot, (int)cell_prims[k], path, a, t_min, best, pend);
                    resolve_pending<F>(pend, a, t_min, tri_base_, best);
                }
                else if (n > kDenseCellMax * (uint32_t)(kDenseRanges - nr)) { // a crowded cell that needs more ranges than are left: it opens the next slice
                    nr = kDenseRanges;
                    continue;
                }
                else {
                    // (a crowded cell - the pole of a UV sphere: 96 triangles meet in a point - takes several ranges of <= kDenseCellMax entries)
                    uint32_t b = beg, left = n;
                    while (left != 0u) {
                        const uint32_t take = left < kDenseCellMax ? left : kDenseCellMax;
                        R.beg0 = nr == 0 ? b : R.beg0, R.beg1 = nr == 1 ? b : R.beg1, R.beg2 = nr == 2 ? b : R.beg2, R.beg3 = nr == 3 ? b : R.beg3;
                        R.cnt |= take << (8 * nr);
                        nr += 1, b += take, left -= take;
                    }
                }
                t_last = tnv[q], ended = endv[q], pos = nextv[q];
                steps_left -= 1;
            }
        }
        // (the ranges filled up inside the batch: the walk takes the cells behind the last one used up again next time)
    }
    ci[0] = (int)(pos & 1023u), ci[1] = (int)((pos >> 10) & 1023u), ci[2] = (int)(pos >> 20);
    R.steps = (uint32_t)(max_cells - steps_left);
    walk_cell = (uint32_t)ci[0] | ((uint32_t)ci[1] << 10) | ((uint32_t)ci[2] << 20);
    walk_t_out = t_out;
    return kWalkGoesOn; // ranges are out (possibly none): test them, then accel_walk_decide()
}
// After the entries of accel_walk_prepare()'s ranges went through test_primitive() / resolve_pending(): is `best` final?
template <typename F> RRTX_DEV int accel_walk_decide(const HitInfo<F> &best, F t_last, F slack_t, bool ended) { return (ended || t_last > best.t + slack_t) ? kWalkDone : kWalkGoesOn; }
// entry j of a lane's ranges -> index into cell_prims (j < the sum of the counts)
RRTX_DEV uint32_t walk_range_entry(uint32_t beg0, uint32_t beg1, uint32_t beg2, uint32_t beg3, uint32_t cnt, uint32_t j)
{
    const uint32_t c0 = cnt & 255u, c1 = (cnt >> 8) & 255u, c2 = (cnt >> 16) & 255u;
    const uint32_t e1 = c0 + c1, e2 = e1 + c2;
    return j < c0 ? beg0 + j : (j < e1 ? beg1 + (j - c0) : (j < e2 ? beg2 + (j - e1) : beg3 + (j - e2)));
}
RRTX_DEV uint32_t walk_range_total(uint32_t cnt) { return (cnt & 255u) + ((cnt >> 8) & 255u) + ((cnt >> 16) & 255u) + (cnt >> 24); }
// The batched walk as ONE lane runs it (the statement tests/path_host_check.cpp holds against the sequential scan):
// prepare a slice, test its entries, decide.
template <typename F, bool SO = false, typename PP, typename HotTab, typename CellTab, typename PrimTab>
RRTX_DEV int accel_closest_hit_batched(const PP &P, const HotTab &hot, const CellTab &cell_start, const PrimTab &cell_prims, const Path<F> &path, F a, F t_min, HitInfo<F> &best, bool resume,
                                       uint32_t &walk_cell, F &walk_t_out, int max_cells, const uint8_t *coarse = nullptr)
{
    WalkRanges R;
    F t_last, slack_t;
    bool ended;
    const int r = accel_walk_prepare<F, SO>(P, hot, cell_start, cell_prims, path, a, t_min, best, resume, walk_cell, walk_t_out, max_cells, R, t_last, slack_t, ended, coarse);
    if (r != kWalkGoesOn) return r;
    PendingRoot<F> pend = {-1, 0, 0};
    const uint32_t n = walk_range_total(R.cnt);
    for (uint32_t j = n; j-- > 0u;) // (backwards: any order must do)
        test_primitive<F, SO>(P, hot, (int)cell_prims[walk_range_entry(R.beg0, R.beg1, R.beg2, R.beg3, R.cnt, j)], path, a, t_min, best, pend);
    resolve_pending<F>(pend, a, t_min, SO ? kNoTriangles : P.n_sph_padded + P.n_msph, best);
    return accel_walk_decide<F>(best, t_last, slack_t, ended);
}

// Per-segment half of the conservative scan filter (see the render kernel's phase 1 and DESIGN.md
// "Conservative scan filter"): with n = d/|d|, u = c.n, s = o.n,
//   disc/|d|^2 = u^2 + 2(o - s n).c + (s^2 - |o|^2) + (r^2 - |c|^2);
// per sphere the host stores c and thr = |c|^2 - r^2 - K eps (|c|^2 + r^2) rounded down, and
//   candidate  <=>  not (u^2 + b.c + g < thr).
// Rays with non-finite or extreme components get g = +inf: everything is a candidate, the exact test decides.
// The filter runs in fp32 whatever F is: an fp64 ray is rounded to float first (kFilterK64 covers that).
struct FilterRay {
    float nx, ny, nz, bx, by, bz, g;
};
RRTX_DEV FilterRay make_filter_ray_k(const Path<float> &path, float a, float K)
{
    FilterRay r = {0, 0, 0, 0, 0, 0, Limits<float>::inf()};
    const float o2 = ffma(path.o.z, path.o.z, ffma(path.o.y, path.o.y, path.o.x * path.o.x));
    if (a >= Limits<float>::tiny() && a <= Limits<float>::big() && o2 <= Limits<float>::big()) {
        const float inv = 1.0f / fsqrt(a);
        r.nx = path.d.x * inv, r.ny = path.d.y * inv, r.nz = path.d.z * inv;
        const float sdot = ffma(path.o.z, r.nz, ffma(path.o.y, r.ny, path.o.x * r.nx));
        r.bx = 2.0f * ffma(-sdot, r.nx, path.o.x);
        r.by = 2.0f * ffma(-sdot, r.ny, path.o.y);
        r.bz = 2.0f * ffma(-sdot, r.nz, path.o.z);
        r.g = ffma(K * 0x1p-24f, o2, ffma(sdot, sdot, -o2)); // K * unit roundoff
    }
    return r;
}
RRTX_DEV FilterRay make_filter_ray_k(const Path<double> &path, double a, float K)
{
    FilterRay r = {0, 0, 0, 0, 0, 0, Limits<float>::inf()};
    // (decided in double: a value that overflows float must not become a finite-looking float)
    const double o2d = path.o.x * path.o.x + path.o.y * path.o.y + path.o.z * path.o.z;
    if (a >= (double)Limits<float>::tiny() && a <= (double)Limits<float>::big() && o2d <= (double)Limits<float>::big()) {
        const float ox = (float)path.o.x, oy = (float)path.o.y, oz = (float)path.o.z, dx = (float)path.d.x, dy = (float)path.d.y, dz = (float)path.d.z;
        const float o2 = ffma(oz, oz, ffma(oy, oy, ox * ox));
        const float inv = 1.0f / fsqrt(ffma(dz, dz, ffma(dy, dy, dx * dx)));
        r.nx = dx * inv, r.ny = dy * inv, r.nz = dz * inv;
        const float sdot = ffma(oz, r.nz, ffma(oy, r.ny, ox * r.nx));
        r.bx = 2.0f * ffma(-sdot, r.nx, ox);
        r.by = 2.0f * ffma(-sdot, r.ny, oy);
        r.bz = 2.0f * ffma(-sdot, r.nz, oz);
        r.g = ffma(K * 0x1p-24f, o2, ffma(sdot, sdot, -o2));
    }
    return r;
}
RRTX_DEV FilterRay make_filter_ray(const Path<float> &path, float a) { return make_filter_ray_k(path, a, (float)kFilterK); }
RRTX_DEV FilterRay make_filter_ray(const Path<double> &path, double a) { return make_filter_ray_k(path, a, (float)kFilterK64); }
// ... with the margins of the filter's form on the matrix cores (rrtx_pack.h: pack_mf_table)
RRTX_DEV FilterRay make_filter_ray_mf(const Path<float> &path, float a) { return make_filter_ray_k(path, a, (float)kFilterKMf); }
RRTX_DEV FilterRay make_filter_ray_mf(const Path<double> &path, double a) { return make_filter_ray_k(path, a, (float)kFilterKMf64); }
RRTX_DEV float filter_value(const FilterRay &r, float cx, float cy, float cz)
{
    const float uu = ffma(cz, r.nz, ffma(cy, r.ny, cx * r.nx));
    const float w = ffma(r.bz, cz, ffma(r.by, cy, ffma(r.bx, cx, r.g)));
    return ffma(uu, uu, w);
}

} // namespace rrtx

#endif
