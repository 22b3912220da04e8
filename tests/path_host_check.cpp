// Host-side check of the accelerated closest hit: compiles the SAME source the
// kernel uses (rrt_amd/csrc/rrtx_path.h: exact tests, tie rules, grid walk; rrtx_grid.h: grid builder)
// for the CPU and compares, ray by ray, the grid walk with the sequential scan of hittable_list.h:95-117.
//   g++ -O2 -std=c++17 -ffp-contract=off tests/path_host_check.cpp -o path_host_check && ./path_host_check
// Scenes: a final.txt-like field of jittered spheres on a huge ground sphere, with big spheres, moving
// spheres, triangles, coincident / nested / lattice-aligned / tiny spheres.  Rays: from the camera
// region, from points on and near the primitives, from inside spheres, from far away (beyond the
// grid's range), axis-parallel and grazing directions.  Exit code 1 on any disagreement.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../rrt_amd/csrc/rrtx_device.h"
#include "../rrt_amd/csrc/rrtx_grid.h"
#include "../rrt_amd/csrc/rrtx_path.h"

using namespace rrtx;

template <typename F> struct Scene {
    std::vector<SphereHot<F>> hot;
    std::vector<SphereCold<F>> cold;
    std::vector<MovingSphereRec<F>> ms;
    std::vector<TriangleRec<F>> tri;
    int n_sph = 0, n_pad = 0;
    CameraRec<F> cam = {};
    KernelParams<F> P = {};
    std::vector<uint32_t> cell_start, always;
    std::vector<GridPrim> cell_prims;
};

template <typename F> static void add_sphere(Scene<F> &s, double x, double y, double z, double r)
{
    SphereHot<F> h;
    h.cx = (F)x, h.cy = (F)y, h.cz = (F)z;
    const F rr = (F)r;
    h.r2 = rr * rr;
    s.hot.push_back(h);
    SphereCold<F> c;
    c.radius = rr, c.mat = 0;
    s.cold.push_back(c);
}

template <typename F> static bool make_scene(Scene<F> &s, int variant, std::mt19937_64 &gen)
{
    std::uniform_real_distribution<double> U(0.0, 1.0);
    add_sphere(s, 0, -1000, 0, 1000);
    const int n = variant == 0 ? 11 : 6;
    for (int a = -n; a < n; ++a)
        for (int b = -n; b < n; ++b) {
            if (variant == 0)
                add_sphere(s, a + 0.9 * U(gen), 0.2, b + 0.9 * U(gen), 0.2);
            else
                add_sphere(s, a, 0.25, b, 0.25); // on the cell lattice
        }
    add_sphere(s, 0, 1, 0, 1.0), add_sphere(s, -4, 1, 0, 1.0), add_sphere(s, 4, 1, 0, 1.0);
    if (variant == 1) {
        add_sphere(s, 0.5, 0.25, 0.5, 0.25), add_sphere(s, 0.5, 0.25, 0.5, 0.25); // coincident: the later one wins
        add_sphere(s, 1.5, 0.3, 1.5, 0.3), add_sphere(s, 1.5, 0.3, 1.5, 0.2);     // nested
        add_sphere(s, -2.5, 0.004, 2.5, 0.004), add_sphere(s, 2.5, 3.0, -2.5, 3.0); // tiny, large
        for (int k = 0; k < 40; ++k) {
            MovingSphereRec<F> m = {};
            const double x = -6 + 12 * U(gen), z = -6 + 12 * U(gen);
            m.c0[0] = (F)x, m.c0[1] = (F)0.2, m.c0[2] = (F)z;
            m.dc[0] = (F)0.3, m.dc[1] = (F)0.3, m.dc[2] = (F)-0.2;
            m.t0 = 0, m.dt = 1, m.radius = (F)0.2, m.r2 = m.radius * m.radius, m.mat = 0;
            s.ms.push_back(m);
        }
        for (int k = 0; k < 4; ++k) {
            TriangleRec<F> t = {};
            const double x = -3 + 6 * U(gen), z = -3 + 6 * U(gen);
            t.v0[0] = (F)x, t.v0[1] = (F)0.9, t.v0[2] = (F)z;
            t.e1[0] = (F)1.5, t.e1[1] = (F)0.2, t.e1[2] = 0;
            t.e2[0] = 0, t.e2[1] = (F)0.1, t.e2[2] = (F)1.5;
            s.tri.push_back(t);
        }
    }
    if (variant == 3) {
        // piles: 300 and 1 300 small spheres crowded around two points - cells of more entries than one range of the batched walk holds
        // (kDenseCellMax: the cell takes several ranges) and than four ranges hold (its lane tests it alone)
        for (int k = 0; k < 300; ++k) add_sphere(s, 2.3 + 0.05 * U(gen), 0.25 + 0.05 * U(gen), -1.7 + 0.05 * U(gen), 0.15 + 0.1 * U(gen));
        for (int k = 0; k < 1300; ++k) add_sphere(s, -3.4 + 0.05 * U(gen), 0.3 + 0.05 * U(gen), 2.6 + 0.05 * U(gen), 0.15 + 0.1 * U(gen));
    }
    if (variant == 2) {
        // a triangle mesh: a UV sphere of radius 1.2 at (0.3, 1.2, -0.4), 24 x 48 quads,
        // and a wavy sheet over a corner of the field
        auto add_tri = [&](const double a[3], const double b[3], const double c[3]) {
            TriangleRec<F> t = {};
            for (int k = 0; k < 3; ++k) t.v0[k] = (F)a[k], t.e1[k] = (F)b[k] - (F)a[k], t.e2[k] = (F)c[k] - (F)a[k];
            const F u1[3] = {t.e1[0], t.e1[1], t.e1[2]}, u2[3] = {t.e2[0], t.e2[1], t.e2[2]};
            t.n[0] = u1[1] * u2[2] - u1[2] * u2[1], t.n[1] = u1[2] * u2[0] - u1[0] * u2[2], t.n[2] = u1[0] * u2[1] - u1[1] * u2[0]; // (direction only: the walk does not use it)
            s.tri.push_back(t);
        };
        const int NU = 24, NV = 48;
        auto sph = [&](int i, int j, double out[3]) {
            const double th = 3.14159265358979 * i / NU, ph = 6.28318530717959 * j / NV;
            out[0] = 0.3 + 1.2 * std::sin(th) * std::cos(ph), out[1] = 1.2 + 1.2 * std::cos(th), out[2] = -0.4 + 1.2 * std::sin(th) * std::sin(ph);
        };
        for (int i = 0; i < NU; ++i)
            for (int j = 0; j < NV; ++j) {
                double p00[3], p01[3], p10[3], p11[3];
                sph(i, j, p00), sph(i, j + 1, p01), sph(i + 1, j, p10), sph(i + 1, j + 1, p11);
                if (i > 0) add_tri(p00, p10, p01);
                if (i < NU - 1) add_tri(p01, p10, p11);
            }
        auto sheet = [&](int i, int j, double out[3]) { out[0] = 3 + 0.1 * i, out[2] = 2 + 0.1 * j, out[1] = 0.6 + 0.15 * std::sin(0.7 * i) * std::cos(0.5 * j); };
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) {
                double p00[3], p01[3], p10[3], p11[3];
                sheet(i, j, p00), sheet(i, j + 1, p01), sheet(i + 1, j, p10), sheet(i + 1, j + 1, p11);
                add_tri(p00, p01, p10), add_tri(p01, p11, p10);
            }
    }
    s.n_sph = (int)s.hot.size();
    s.n_pad = (s.n_sph + kSpherePad - 1) / kSpherePad * kSpherePad;
    while ((int)s.hot.size() < s.n_pad) { // never-hit padding, as the library packs it
        SphereHot<F> h = {0, 0, 0, -std::numeric_limits<F>::infinity()};
        s.hot.push_back(h);
        SphereCold<F> c = {1, 0};
        s.cold.push_back(c);
    }
    s.cam.origin[0] = 13, s.cam.origin[1] = 2, s.cam.origin[2] = 3;
    s.cam.time0 = 0, s.cam.time1 = variant == 1 ? (F)1 : (F)0;
    if (s.ms.empty()) s.ms.resize(1);
    if (s.tri.empty()) s.tri.resize(1);
    const int n_ms = variant == 1 ? 40 : 0, n_tri = variant == 1 ? 4 : (variant == 2 ? (int)s.tri.size() : 0);
    GridRec<F> G = {};
    if (!build_grid<F>(s.hot, s.cold, s.n_sph, s.n_pad, s.ms, n_ms, s.tri, n_tri, s.cam, s.cell_start, s.cell_prims, s.always, G)) return false;
    if (s.cell_prims.empty()) s.cell_prims.push_back(0);
    KernelParams<F> &P = s.P;
    P.sph_hot = s.hot.data(), P.sph_cold = s.cold.data(), P.msph = s.ms.data(), P.tri = s.tri.data();
    P.n_sph = s.n_sph, P.n_sph_padded = s.n_pad, P.n_msph = n_ms, P.n_tri = n_tri;
    P.grid = G, P.grid_cell_start = s.cell_start.data(), P.grid_cell_prims = s.cell_prims.data(), P.grid_always = s.always.empty() ? nullptr : s.always.data();
    P.n_always = (int)s.always.size(), P.n_grid_cells = (int)s.P.grid.dims[0] * (int)s.P.grid.dims[1] * (int)s.P.grid.dims[2], P.n_grid_prims = (int)s.cell_start[(size_t)P.n_grid_cells];
    return true;
}

template <typename F> static HitInfo<F> sequential(const Scene<F> &s, const Path<F> &path, F a, F t_min)
{
    const KernelParams<F> &P = s.P;
    HitInfo<F> best = {std::numeric_limits<F>::infinity(), -1};
    for (int q = 0; q < P.n_sph; ++q) refine_sphere<F>(s.hot[q].cx, s.hot[q].cy, s.hot[q].cz, s.hot[q].r2, path, a, t_min, q, best);
    for (int q = 0; q < P.n_msph; ++q) {
        const V3<F> cen = msphere_center<F>(s.ms[q], path.tm);
        refine_sphere<F>(cen.x, cen.y, cen.z, s.ms[q].r2, path, a, t_min, P.n_sph_padded + q, best);
    }
    for (int q = 0; q < P.n_tri; ++q) {
        F tt;
        if (triangle_test<F, true>(s.tri[q], path, t_min, best.t, tt)) best.t = tt, best.idx = P.n_sph_padded + P.n_msph + q;
    }
    return best;
}

template <typename F> static int run(const char *name, int variant, long n_rays, uint64_t seed)
{
    std::mt19937_64 gen(seed);
    Scene<F> s;
    if (!make_scene<F>(s, variant, gen)) {
        std::printf("%s variant %d: no grid built\n", name, variant);
        return 1;
    }
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    long mismatches = 0, walked = 0, scanned = 0, sliced_diff = 0, batched_diff = 0;
    const F t_min = (F)0.001;
    for (long i = 0; i < n_rays; ++i) {
        Path<F> path = {};
        const int kind = (int)(i % 8);
        double o[3], d[3] = {N(gen), N(gen), N(gen)};
        if (kind == 0) { // camera region, looking at the scene
            o[0] = 13 + 0.1 * N(gen), o[1] = 2 + 0.1 * N(gen), o[2] = 3 + 0.1 * N(gen);
            d[0] = -13 + 8 * (U(gen) - 0.5), d[1] = -2 + 2 * (U(gen) - 0.5), d[2] = -3 + 8 * (U(gen) - 0.5);
        }
        else if (kind == 1 || kind == 2) { // on the ground among the spheres, going anywhere upward / grazing
            o[0] = 24 * (U(gen) - 0.5), o[2] = 24 * (U(gen) - 0.5), o[1] = std::sqrt(1e6 - o[0] * o[0] - o[2] * o[2]) - 1000.0;
            d[1] = kind == 1 ? std::fabs(d[1]) : 0.02 * d[1];
        }
        else if (kind == 3) { // on / inside a small sphere
            const int q = 1 + (int)(U(gen) * (s.n_sph - 1));
            const double r = std::sqrt((double)s.hot[q].r2) * (U(gen) < 0.5 ? 1.0 : U(gen));
            double u[3] = {N(gen), N(gen), N(gen)}, l = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            o[0] = s.hot[q].cx + r * u[0] / l, o[1] = s.hot[q].cy + r * u[1] / l, o[2] = s.hot[q].cz + r * u[2] / l;
        }
        else if (kind == 4) { // axis-parallel and lattice-aligned
            o[0] = std::floor(12 * (U(gen) - 0.5)) + (U(gen) < 0.5 ? 0.0 : 0.5), o[1] = 0.25, o[2] = std::floor(12 * (U(gen) - 0.5));
            d[0] = U(gen) < 0.5 ? 1 : 0, d[1] = 0, d[2] = d[0] == 0 ? -1 : (U(gen) < 0.5 ? 0 : 1);
        }
        else if (kind == 5) { // far away on the ground, heading back towards the scene or anywhere
            const double ang = 6.283185307 * U(gen), dist = 60 + 900 * U(gen);
            o[0] = dist * std::cos(ang), o[2] = dist * std::sin(ang), o[1] = std::sqrt(1e6 - dist * dist) - 1000.0;
            if (U(gen) < 0.7) d[0] = -o[0] + 10 * N(gen), d[1] = -o[1] + 0.5 * U(gen), d[2] = -o[2] + 10 * N(gen);
        }
        else if (kind == 6) { // high above, looking down
            o[0] = 30 * (U(gen) - 0.5), o[1] = 5 + 40 * U(gen), o[2] = 30 * (U(gen) - 0.5);
            d[1] = -std::fabs(d[1]) - 1;
        }
        else if (kind == 7 && s.P.n_tri > 4 && (i & 8)) { // in the plane of a triangle, up to a perturbation of 1e-12 .. 1e-3: where Moeller-Trumbore is at its worst
            const TriangleRec<F> &t = s.tri[(size_t)(U(gen) * s.P.n_tri) % (size_t)s.P.n_tri];
            const double bu = 3 * U(gen) - 1, bv = 3 * U(gen) - 1, cu = 6 * (U(gen) - 0.5), cv = 6 * (U(gen) - 0.5), pert = std::pow(10.0, -12 + 9 * U(gen));
            for (int k = 0; k < 3; ++k) {
                const double target = (double)t.v0[k] + bu * (double)t.e1[k] + bv * (double)t.e2[k];
                o[k] = (double)t.v0[k] + cu * 3 * (double)t.e1[k] + cv * 3 * (double)t.e2[k];
                d[k] = target - o[k] + pert * N(gen);
            }
        }
        else { // anywhere near, any direction, any scale
            o[0] = 40 * (U(gen) - 0.5), o[1] = 3 * U(gen), o[2] = 40 * (U(gen) - 0.5);
            const double sc = std::pow(10.0, 4 * (U(gen) - 0.5));
            d[0] *= sc, d[1] *= sc, d[2] *= sc;
        }
        path.o = mk<F>((F)o[0], (F)o[1], (F)o[2]), path.d = mk<F>((F)d[0], (F)d[1], (F)d[2]);
        path.tm = (F)(s.cam.time0 + (s.cam.time1 - s.cam.time0) * U(gen));
        const F a = vlen2<F>(path.d);
        if (!(a > 0)) continue;
        const HitInfo<F> want = sequential<F>(s, path, a, t_min);
        // the walk in one go ...
        HitInfo<F> best = {std::numeric_limits<F>::infinity(), -1};
        uint32_t cell = 0;
        F t_out = 0;
        int r = accel_closest_hit<F>(s.P, s.hot.data(), s.cell_start.data(), s.cell_prims.data(), path, a, t_min, best, false, cell, t_out, s.P.grid.max_steps);
        if (r == kWalkNeedsScan || r == kWalkFarScan) {
            scanned += 1;
            continue;
        }
        walked += 1;
        if (r != kWalkDone || best.idx != want.idx || !(best.t == want.t)) {
            if (mismatches < 10)
                std::printf("  mismatch (kind %d): walk %d %.9g, scan %d %.9g, o %.9g %.9g %.9g d %.9g %.9g %.9g\n", kind, best.idx, (double)best.t, want.idx, (double)want.t, o[0], o[1], o[2], d[0],
                            d[1], d[2]);
            mismatches += 1;
        }
        // ... and in slices of 1 .. 4 cells, as the kernel does
        HitInfo<F> b2 = {std::numeric_limits<F>::infinity(), -1};
        bool resume = false;
        const int slice = 1 + (int)(i % 4);
        for (int guard = 0; guard < 100000; ++guard) {
            r = accel_closest_hit<F>(s.P, s.hot.data(), s.cell_start.data(), s.cell_prims.data(), path, a, t_min, b2, resume, cell, t_out, slice);
            if (r != kWalkGoesOn) break;
            resume = true;
        }
        if (r != kWalkDone || b2.idx != want.idx || !(b2.t == want.t)) sliced_diff += 1;
        // ... and the batched walk (the render kernel's: a slice's cells are listed first, their entries tested afterwards in any
        // order - here backwards -, then the decision): same answer, same slice boundaries
        HitInfo<F> b3 = {std::numeric_limits<F>::infinity(), -1};
        resume = false;
        // (with the empty-block skipping where the grid carries the coarse occupancy bytes - built with -DRRTX_GRID_COARSE_ALWAYS=1:
        // every grid here -, in slices of 1 .. 16 steps)
        const uint8_t *coarse = s.P.grid.coarse_off ? (const uint8_t *)(s.cell_start.data() + s.P.grid.coarse_off) : nullptr;
        const int batched_slice = 1 + (int)(i % 16);
        for (int guard = 0; guard < 100000; ++guard) {
            r = accel_closest_hit_batched<F>(s.P, s.hot.data(), s.cell_start.data(), s.cell_prims.data(), path, a, t_min, b3, resume, cell, t_out, batched_slice, coarse);
            if (r != kWalkGoesOn) break;
            resume = true;
        }
        if (r != kWalkDone || b3.idx != want.idx || !(b3.t == want.t)) batched_diff += 1;
    }
    std::printf("%s variant %d: grid %d x %d x %d, %d entries, %d always, far %.4g | %ld rays walked, %ld left to the scan, %ld mismatches, %ld sliced-walk mismatches, %ld batched-walk mismatches\n", name, variant,
                s.P.grid.dims[0], s.P.grid.dims[1], s.P.grid.dims[2], s.P.n_grid_prims, s.P.n_always, std::sqrt((double)s.P.grid.far2), walked, scanned, mismatches, sliced_diff, batched_diff);
    return mismatches != 0 || sliced_diff != 0 || batched_diff != 0 || walked == 0;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 400000;
    int bad = 0;
    bad |= run<float>("fp32", 0, n, 1);
    bad |= run<float>("fp32", 1, n, 2);
    bad |= run<double>("fp64", 0, n / 2, 3);
    bad |= run<double>("fp64", 1, n / 2, 4);
    bad |= run<double>("fp64", 2, n / 2, 5); // the mesh: gridded in fp64 ...
    bad |= run<float>("fp32", 3, n / 4, 7), bad |= run<double>("fp64", 3, n / 8, 8); // piles: crowded cells
    {
        std::mt19937_64 gen(6);
        Scene<float> s; // ... and not in fp32, where the bound admits no triangle of this size: no grid, the list is scanned
        if (make_scene<float>(s, 2, gen)) std::printf("fp32 variant 2: a grid was built for a mesh the bound does not cover\n"), bad |= 1;
    }
    return bad;
}
