// Host side of the accelerated closest hit: builds the uniform grid + always-list that
// accel_closest_hit (rrtx_path.h) walks.  Header-only so that tests/path_host_check.cpp can build grids
// and walk them on the CPU with the very code the library and the kernel use.
#ifndef RRTX_GRID_H
#define RRTX_GRID_H

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <cstring>
#include <vector>

#include "rrtx_device.h"

#ifndef RRTX_GRID_COARSE_ALWAYS // 1: every grid gets the coarse occupancy bytes (tests/path_host_check.cpp: the skipping on small grids too)
#define RRTX_GRID_COARSE_ALWAYS 0
#endif
#ifndef RRTX_GRID_SLICE_OVERRIDE
#define RRTX_GRID_SLICE_OVERRIDE 0
#endif

namespace rrtx {

// Tuning knobs read from the environment exist in experiment builds only (-DRRTX_EXPERIMENTS): a stray
// variable must not change the grid of the shipped drop-in.
inline double grid_knob(const char *name, double dflt)
{
#ifdef RRTX_EXPERIMENTS
    if (const char *v = getenv(name)) return atof(v);
#else
    (void)name;
#endif
    return dflt;
}

// ---------------------------------------------------------------------------------------------
// Acceleration grid for use_bvh != 0 (device side: accel_closest_hit).
//
// Spheres and moving spheres of ordinary size go into a uniform grid (cell = 2 x their median
// extent; measured on final.txt: 1.5 / 2 / 2.5 / 3.5 -> 53.2 / 49.5 / 51.5 / 74 ms at spp 504); primitives
// larger than 1.6 cells (RRTX_GRID_LARGE; with the three r = 1 spheres of final.txt gridded the grid has
// three layers of cells instead of one: 66.8 against 41.1 ms - the factor is only raised, to 4 and 10, when
// the always-list would overflow), spheres smaller than a fiftieth of a cell and the triangles the bound
// below does not cover go into the "always" list.  Every gridded primitive is entered into all cells its box, INFLATED, overlaps.
// By how much: the reference's discriminant, evaluated in floating point, can be >= 0 only if the
// ray's line passes within sqrt(r^2 + m) of the centre, m = 32 eps (|o - c|^2 + r^2) (a bound on the
// rounding error of (oc.d)^2 - |d|^2 (|oc|^2 - r^2) relative to |d|^2; 24 eps by the usual gamma_n
// accounting), and the root it then reports lies within sqrt(m) of the sphere along the ray.  The grid
// answers rays that start within `far` of its centre (six half diagonals, three if that blows a
// typical box up by more than a tenth of a cell; always past the camera), so primitive p gets
// delta_p = sqrt(r_p^2 + m_p) - r_p + 0.01 cell with |o - c| <= far + half diagonal in m_p: small
// spheres grow more than large ones, and what would grow by more than half a cell joins the always
// list.  A primitive whose test can succeed is then always found in a cell the ray's exact line
// passes through or within delta_p of — and a 3-D DDA that is off by a few ulps at a cell boundary only
// ever trades a cell the ray grazes by less than that for its neighbour.  Rays starting further away
// are tested against the grid's box blown up by their own sqrt(m) and, if they touch it, scanned.
// The walk stops once the next cell begins more than slack0 + slack1 (|o - centre| + half diagonal)
// beyond the closest hit, slack1 = 1.5 sqrt(32 eps): the along-ray error of a root.
//
// Triangles.  The reference's test (triangle.h:38-75) is Moeller-Trumbore with the
// absolute cut |a| < 1e-7, a = e1 . (d x e2).  Its computed a, and the numerators of t, u, v, carry
// rounding errors of at most 16 eps |d| |e1| |e2| resp. 16 eps |s| |e1| |e2|, |s| |d| |e2|, |s| |d| |e1|
// (s = o - v0; cross product + dot product, gamma_n accounting with room to spare).  With
// rho = 16 eps |d| |e1| |e2| / 1e-7 < 1 an accepted hit therefore has |a_exact| >= (1 - rho) 1e-7, and
// inserting the computed (t, u, v) into o + t d = v0 + u e1 + v e2 leaves a residual of at most
//     R = rho / (1 - rho) (3 |s| + t |d| + |e1| + |e2|):
// the reported hit point lies within R of the triangle (u, v in [0, 1]).  A triangle is gridded - entered
// in every cell its box, inflated by R + a hundredth of a cell, overlaps - if rho <= 1e-4 for every ray
// the walk answers (|d|^2 <= dir2_max = 1e6, |s| and t |d| within far + the grid's half diagonal), the
// inflation stays under half a cell and its box under `large` cells; everything else stays in the
// always-list.  In fp64 rho ~ 2e-8 |d| |e1| |e2|: meshes are gridded.  In fp32 rho ~ 10 |d| |e1| |e2|: the
// bound admits nothing of practical size, and a mesh scene keeps the list scan.
// ---------------------------------------------------------------------------------------------
#ifndef RRTX_GRID_TRI_SAT // 1: a triangle is listed only in the cells it reaches (triangle_touches_box), 0: in every cell of its box
#define RRTX_GRID_TRI_SAT 1
#endif
#ifndef RRTX_APPROX_TRI_INFLATION
#define RRTX_APPROX_TRI_INFLATION 0.05
#endif
constexpr double kApproxTriInflation = RRTX_APPROX_TRI_INFLATION; // of a cell: boxes of triangles gridded without proof (fp32)
constexpr double kGridDir2Max = 1e6; // |d|^2 up to which gridded triangles are proven (camera rays of final.txt: ~ 1e2)
// Does the triangle (v0, v1, v2) touch the axis-aligned box [lo, hi]?  The separating-axis test (Akenine-Moeller 2001: the
// box's three face normals, the triangle's normal, the nine cross products of edges and axes), in double, told to answer
// "yes" whenever an axis separates by less than `margin`.  Used to keep a triangle out of the cells of its bounding box
// that it does not reach: a mesh of cell-sized triangles lists each in ~8 cells by its box, in ~4 by this test.
inline bool triangle_touches_box(const double v[3][3], const double lo[3], const double hi[3], double margin)
{
    double c[3], h[3], p[3][3];
    for (int k = 0; k < 3; ++k) c[k] = 0.5 * (lo[k] + hi[k]), h[k] = 0.5 * (hi[k] - lo[k]) + margin;
    for (int q = 0; q < 3; ++q)
        for (int k = 0; k < 3; ++k) p[q][k] = v[q][k] - c[k];
    for (int k = 0; k < 3; ++k) { // the box's face normals
        const double mn = std::min({p[0][k], p[1][k], p[2][k]}), mx = std::max({p[0][k], p[1][k], p[2][k]});
        if (mn > h[k] || mx < -h[k]) return false;
    }
    const double e[3][3] = {{p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]}, {p[2][0] - p[1][0], p[2][1] - p[1][1], p[2][2] - p[1][2]}, {p[0][0] - p[2][0], p[0][1] - p[2][1], p[0][2] - p[2][2]}};
    auto separated = [&](const double a[3]) { // is `a` a separating axis?  (a zero axis - parallel edge and axis - separates nothing)
        const double len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (!(len > 0)) return false;
        const double d0 = a[0] * p[0][0] + a[1] * p[0][1] + a[2] * p[0][2], d1 = a[0] * p[1][0] + a[1] * p[1][1] + a[2] * p[1][2], d2 = a[0] * p[2][0] + a[1] * p[2][1] + a[2] * p[2][2];
        const double r = h[0] * std::fabs(a[0]) + h[1] * std::fabs(a[1]) + h[2] * std::fabs(a[2]);
        return std::min({d0, d1, d2}) > r || std::max({d0, d1, d2}) < -r;
    };
    for (int q = 0; q < 3; ++q)
        for (int k = 0; k < 3; ++k) {
            double a[3] = {0, 0, 0}; // axis_k x e_q
            const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            a[k1] = -e[q][k2], a[k2] = e[q][k1];
            if (separated(a)) return false;
        }
    const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    return !separated(n);
}

template <typename F>
inline bool build_grid(const std::vector<SphereHot<F>> &hot, const std::vector<SphereCold<F>> &cold, int n_sph, int n_sph_pad, const std::vector<MovingSphereRec<F>> &ms,
                       int n_msph, const std::vector<TriangleRec<F>> &tri, int n_tri, const CameraRec<F> &cam, std::vector<uint32_t> &cell_start, std::vector<GridPrim> &cell_prims,
                       std::vector<uint32_t> &always, GridRec<F> &G, bool allow_approximate = false, bool *approximate = nullptr)
{
    if (approximate) *approximate = false;
    const int msph_base = n_sph_pad, tri_base = n_sph_pad + n_msph;
    const double eps0 = sizeof(F) == 4 ? 0x1p-24 : 0x1p-53;
    if ((int64_t)tri_base + n_tri >= (1 << 28)) return false;
    struct Box {
        double lo[3], hi[3], r;
        int idx;
        bool is_tri;
        double e1e2, e_sum, vmax; // triangles: |e1| |e2|, |e1| + |e2|, largest vertex coordinate
        double v[3][3];           // triangles: the three vertices the device test sees
        bool candidate;           // may be gridded at all (triangles: only where the bound holds, or under the approximate rule)
        bool proven;              // triangles: the residual bound holds
    };
    std::vector<Box> boxes;
    for (int i = 0; i < n_sph; ++i) {
        Box b;
        const double cc[3] = {(double)hot[i].cx, (double)hot[i].cy, (double)hot[i].cz}, r = std::fabs((double)cold[i].radius);
        for (int k = 0; k < 3; ++k) b.lo[k] = cc[k] - r, b.hi[k] = cc[k] + r;
        b.r = r, b.idx = i, b.is_tri = false, b.candidate = true, b.proven = true;
        boxes.push_back(b);
    }
    for (int i = 0; i < n_msph; ++i) {
        // centre(tm) = c0 + ((tm - t0) / dt) dc is linear in tm: the hull over the shutter interval is the
        // hull of its end points (the ray's time is drawn from [time0, time1], camera.h:37)
        Box b;
        const double r = std::fabs((double)ms[i].radius);
        for (int k = 0; k < 3; ++k) b.lo[k] = 1e300, b.hi[k] = -1e300;
        for (double tm : {(double)cam.time0, (double)cam.time1}) {
            const double sfrac = (tm - (double)ms[i].t0) / (double)ms[i].dt;
            for (int k = 0; k < 3; ++k) {
                const double ck = (double)ms[i].c0[k] + sfrac * (double)ms[i].dc[k];
                const double pad = 1e-5 * (std::fabs(ck) + r); // the device evaluates the centre in F
                b.lo[k] = std::min(b.lo[k], ck - r - pad), b.hi[k] = std::max(b.hi[k], ck + r + pad);
            }
        }
        b.r = r, b.idx = msph_base + i, b.is_tri = false, b.candidate = true, b.proven = true;
        boxes.push_back(b);
    }
    for (int i = 0; i < n_tri; ++i) {
        // the vertices the device test sees: v0, v0 + e1, v0 + e2 with the stored (rounded) edges
        Box b;
        double l1 = 0, l2 = 0;
        b.vmax = 0;
        for (int k = 0; k < 3; ++k) {
            const double a0 = (double)tri[i].v0[k], a1 = a0 + (double)tri[i].e1[k], a2 = a0 + (double)tri[i].e2[k];
            b.lo[k] = std::min({a0, a1, a2}), b.hi[k] = std::max({a0, a1, a2});
            b.v[0][k] = a0, b.v[1][k] = a1, b.v[2][k] = a2;
            l1 += (double)tri[i].e1[k] * (double)tri[i].e1[k], l2 += (double)tri[i].e2[k] * (double)tri[i].e2[k];
            b.vmax = std::max({b.vmax, std::fabs(a0), std::fabs(a1), std::fabs(a2)});
        }
        b.e1e2 = std::sqrt(l1) * std::sqrt(l2), b.e_sum = std::sqrt(l1) + std::sqrt(l2);
        b.r = 1e300, b.idx = tri_base + i, b.is_tri = true; // (r: never "tiny")
        b.proven = 16 * eps0 * std::sqrt(kGridDir2Max) * b.e1e2 / 1e-7 <= 1e-4; // rho, see above
        b.candidate = b.proven || allow_approximate;
        boxes.push_back(b);
    }
    std::vector<double> ext, sorted;
    for (const Box &b : boxes) {
        ext.push_back(std::max({b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]}));
        if (b.candidate) sorted.push_back(ext.back());
    }
    if (sorted.size() < 32) return false; // nothing to gain on a handful of primitives
    std::nth_element(sorted.begin(), sorted.begin() + sorted.size() / 2, sorted.end());
    double cell = grid_knob("RRTX_GRID_CELL", 2.0) * sorted[sorted.size() / 2];
    if (!(cell > 0) || !std::isfinite(cell)) return false;

    const double large0 = grid_knob("RRTX_GRID_LARGE", 1.6);
    double large = large0;
    const double eps = sizeof(F) == 4 ? 0x1p-24 : 0x1p-53;
    const double cam_o[3] = {(double)cam.origin[0], (double)cam.origin[1], (double)cam.origin[2]};
    std::vector<double> delta(boxes.size(), 0.0); // per primitive: how far its box is inflated
    for (int attempt = 0; attempt < 48; ++attempt) {
        // half diagonal and centre of what would be gridded (by size alone: the inflation comes next)
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        size_t n_sized = 0;
        for (size_t i = 0; i < boxes.size(); ++i)
            if (boxes[i].candidate && !(ext[i] > large * cell || boxes[i].r < cell / 50)) {
                n_sized += 1;
                for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], boxes[i].lo[k]), hi[k] = std::max(hi[k], boxes[i].hi[k]);
            }
        if (n_sized < 32) return false;
        double hd = 0.5 * std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2])) + 0.5 * cell;
        double cam_dist = 0;
        for (int k = 0; k < 3; ++k) cam_dist += (cam_o[k] - 0.5 * (lo[k] + hi[k])) * (cam_o[k] - 0.5 * (lo[k] + hi[k]));
        cam_dist = std::sqrt(cam_dist);
        // The grid answers rays that start within `far` of its centre: the larger, the fewer rays need the
        // fat-box test, but the more every box must be inflated.  Six half diagonals if that keeps the
        // inflation of a typical primitive under a tenth of a cell, else three; always past the camera.
        double far = 0, rho_max = 0;
        bool built = false, any_approximate = false;
        for (double mult : {6.0, 3.0}) {
            far = std::max(mult * hd, 1.25 * cam_dist);
            const double far_cap = sizeof(F) == 4 ? 1e6 : 1e50;
            if (far > far_cap) far = far_cap;
            const double R = far + hd; // |o - c| of any ray the grid answers
            always.clear();
            std::vector<double> infl;
            rho_max = 0;
            any_approximate = false;
            for (size_t i = 0; i < boxes.size(); ++i) {
                bool grid_it;
                if (boxes[i].is_tri && boxes[i].proven) {
                    const double rho = 16 * eps * std::sqrt(kGridDir2Max) * boxes[i].e1e2 / 1e-7;
                    const double resid = rho < 0.5 ? rho / (1 - rho) * (4 * R + boxes[i].e_sum) : 1e300;
                    delta[i] = resid + 4 * eps * boxes[i].vmax + 0.01 * cell;
                    grid_it = !(ext[i] > large * cell) && !(delta[i] > 0.5 * cell) && std::isfinite(delta[i]);
                    if (grid_it) rho_max = std::max(rho_max, rho);
                }
                else if (boxes[i].is_tri) {
                    // The approximate rule (fp32 meshes, allow_approximate): the bound above admits nothing there - a ray that grazes
                    // a triangle's plane (|a| between the reference's cut of 1e-7 and ~1e-4 |d| |e1| |e2|) can be reported to hit
                    // it from units away - but such pairs are rare (measured: rrtx_stats.list_mismatches of the VERIFY build,
                    // tests/test_gpu_mesh.py), and the reference's own BVH misses them against its own list scan just the same
                    // (bvh.h:167-175: boxes without any inflation).  kApproxTriInflation of a cell on every side.
                    delta[i] = kApproxTriInflation * cell + 4 * eps * boxes[i].vmax;
                    grid_it = boxes[i].candidate && !(ext[i] > large * cell) && std::isfinite(delta[i]);
                    if (grid_it) any_approximate = true;
                }
                else {
                    const double r = boxes[i].r, m = 32 * eps * (R * R + r * r);
                    delta[i] = std::sqrt(r * r + m) - r + 0.01 * cell; // the test can succeed up to sqrt(r^2 + m) from the centre
                    grid_it = !(ext[i] > large * cell || boxes[i].r < cell / 50 || delta[i] > 0.5 * cell);
                }
                if (grid_it)
                    infl.push_back(delta[i]);
                else
                    always.push_back((uint32_t)boxes[i].idx);
            }
            if (always.size() > 48 || infl.size() < 32) continue;
            std::nth_element(infl.begin(), infl.begin() + infl.size() / 2, infl.end());
            if (mult > 3.0 && infl[infl.size() / 2] > 0.1 * cell) continue;
            built = true;
            break;
        }
        if (!built) {
            // too many primitives left over for the always-list: first let larger ones into the cells (a mesh of
            // small triangles among spheres ten times their size: 32 against 87 ms with the cells kept small),
            // then try larger cells, where "large" primitives become ordinary and inflations relatively smaller
            if (large < 9.9)
                large *= 2.5;
            else
                large = large0, cell *= 1.6;
            continue;
        }
        std::vector<int> gridded;
        {
            std::vector<char> is_always(boxes.size(), 0);
            size_t ai = 0; // always[] was filled in box order
            for (size_t i = 0; i < boxes.size(); ++i)
                if (ai < always.size() && always[ai] == (uint32_t)boxes[i].idx) is_always[i] = 1, ++ai;
            for (size_t i = 0; i < boxes.size(); ++i)
                if (!is_always[i]) gridded.push_back((int)i);
        }
        for (int k = 0; k < 3; ++k) lo[k] = 1e300, hi[k] = -1e300;
        double slack_max = 0;
        for (int i : gridded) {
            slack_max = std::max(slack_max, delta[i]);
            for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], boxes[i].lo[k] - delta[i]), hi[k] = std::max(hi[k], boxes[i].hi[k] + delta[i]);
        }
        int dims[3];
        double total = 1;
        for (int k = 0; k < 3; ++k) {
            dims[k] = (int)std::ceil((hi[k] - lo[k]) / cell);
            if (dims[k] < 1) dims[k] = 1;
            total *= dims[k];
        }
        // (2 M cells = 8 MB of cell_start: with 262 144 a mesh of 27 k triangles had to make do with cells of 0.14
        // instead of 0.087, 31.7 against 22.3 ms)
        if (total > grid_knob("RRTX_GRID_MAXCELLS", 2097152.0) || dims[0] > 1023 || dims[1] > 1023 || dims[2] > 1023) { // (the kernel packs a cell's coordinates into 3 x 10 bits)
            large = large0, cell *= 1.6;
            continue;
        }
        hd = 0.5 * std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        // fill
        const int ncell = dims[0] * dims[1] * dims[2];
        std::vector<uint32_t> count(ncell + 1, 0);
        auto range = [&](int i, int k, int &a, int &z) {
            a = (int)std::floor((boxes[i].lo[k] - delta[i] - lo[k]) / cell), z = (int)std::floor((boxes[i].hi[k] + delta[i] - lo[k]) / cell);
            a = std::max(a, 0), z = std::min(z, dims[k] - 1);
        };
        for (int pass = 0; pass < 2; ++pass) {
            for (int i : gridded) {
                int a[3], z[3];
                for (int k = 0; k < 3; ++k) range(i, k, a[k], z[k]);
                for (int iz = a[2]; iz <= z[2]; ++iz)
                    for (int iy = a[1]; iy <= z[1]; ++iy)
                        for (int ix = a[0]; ix <= z[0]; ++ix) {
                            const int cidx = (iz * dims[1] + iy) * dims[0] + ix;
                            if (boxes[i].is_tri && RRTX_GRID_TRI_SAT) {
                                // (a hit the exact test reports lies within delta of the triangle: a point of the triangle
                                // then lies in the hit's cell grown by delta on every side)
                                const int ic[3] = {ix, iy, iz};
                                double blo[3], bhi[3];
                                for (int k = 0; k < 3; ++k) blo[k] = lo[k] + ic[k] * cell - delta[i], bhi[k] = lo[k] + (ic[k] + 1) * cell + delta[i];
                                if (!triangle_touches_box(boxes[i].v, blo, bhi, 1e-6 * cell)) continue;
                            }
                            if (pass == 0)
                                count[cidx + 1] += 1;
                            else
                                cell_prims[count[cidx]++] = (GridPrim)boxes[i].idx;
                        }
            }
            if (pass == 0) {
                for (int q = 0; q < ncell; ++q) count[q + 1] += count[q];
                cell_start.assign(count.begin(), count.end());
                cell_prims.assign(count[ncell], 0);
                if (count[ncell] > 4000000u) return false;
            }
        }
        // within a cell keep primitive order (not needed for correctness; keeps runs deterministic)
        for (int q = 0; q < ncell; ++q) std::sort(cell_prims.begin() + cell_start[q], cell_prims.begin() + cell_start[q + 1]);
        for (int k = 0; k < 3; ++k) {
            G.gmin[k] = (F)lo[k], G.gmax[k] = (F)(lo[k] + dims[k] * cell);
            G.cell[k] = (F)cell, G.inv_cell[k] = (F)(1.0 / cell);
            G.dims[k] = dims[k];
            G.center[k] = (F)(0.5 * (lo[k] + hi[k]));
        }
        G.far2 = (F)(far * far);
        G.slack = (F)(0.01 * cell);              // slack0
        G.slack1 = (F)std::max(1.5 * std::sqrt(32 * eps), 8 * rho_max); // times (|o - centre| + half diagonal); 8 rho: a gridded triangle's residual
        G.dir2_max = (F)(rho_max > 0 || any_approximate ? kGridDir2Max : (sizeof(F) == 4 ? 1e15 : 1e120)); // (Limits<F>::coop_big() when no triangle is gridded)
        if (approximate) *approximate = any_approximate;
        G.half_diag = (F)hd;
        G.max_steps = dims[0] + dims[1] + dims[2] + 3 + (int)cell_prims.size(); // trips of the walk: a cell step or a primitive test each
        // slices of the walk: short in a dense grid, where walks are short (final.txt, one layer of 29 x 29 cells: 4 is
        // the measured optimum - 2 / 3 / 6 cost 15 / 6 / 8 % -; 40 000 spheres in one layer: 8.3 ms with 4, 10.3 with
        // 16), long where a ray crosses dozens of empty cells (a 118 x 23 x 115 mesh grid, 75 % of it empty: 23.6 ms
        // with 4, 20.1 with 8, 19.7 with 16; the same mesh at a ninth of the triangles, 40 x 8 x 39 cells: 11.4 with 4,
        // 13.0 with 16)
        {
            size_t empty = 0;
            for (int q = 0; q < ncell; ++q) empty += cell_start[q + 1] == cell_start[q];
            G.walk_slice = RRTX_GRID_SLICE_OVERRIDE > 0 ? RRTX_GRID_SLICE_OVERRIDE : (2 * empty > (size_t)ncell && dims[0] + dims[1] + dims[2] >= 128 ? 16 : 4);
        }
        // empty-space skipping for the grids that got long slices (see rrtx_device.h, GridRec::coarse_off)
        G.coarse_off = 0, G.coarse_dims[0] = G.coarse_dims[1] = G.coarse_dims[2] = 0;
        if (RRTX_GRID_COARSE_ALWAYS || (G.walk_slice > 4 && grid_knob("RRTX_GRID_COARSE", 0.0) != 0.0)) { // (off in the product: measured slower, rrtx_kernels.hip)
            const int cd[3] = {(dims[0] + 3) / 4, (dims[1] + 3) / 4, (dims[2] + 3) / 4};
            std::vector<uint8_t> occ((size_t)cd[0] * cd[1] * cd[2], 0);
            for (int iz = 0; iz < dims[2]; ++iz)
                for (int iy = 0; iy < dims[1]; ++iy)
                    for (int ix = 0; ix < dims[0]; ++ix) {
                        const int q = (iz * dims[1] + iy) * dims[0] + ix;
                        if (cell_start[q + 1] != cell_start[q]) occ[((size_t)(iz / 4) * cd[1] + iy / 4) * cd[0] + ix / 4] = 1;
                    }
            G.coarse_off = ncell + 1;
            for (int k = 0; k < 3; ++k) G.coarse_dims[k] = cd[k];
            const size_t words = (occ.size() + 3) / 4;
            const size_t at = cell_start.size();
            cell_start.resize(at + words, 0u);
            memcpy(&cell_start[at], occ.data(), occ.size());
        }
        if (grid_knob("RRTX_DEBUG_GRID", 0.0) != 0.0)
            fprintf(stderr, "rrtx grid: cell %g dims %d x %d x %d, %zu entries, %zu always, largest inflation %g, half diagonal %g, far %g, centre %g %g %g\n", cell, dims[0], dims[1], dims[2],
                    cell_prims.size(), always.size(), slack_max, hd, far, (double)G.center[0], (double)G.center[1], (double)G.center[2]);
        return true;
    }
    return false;
}

// A grid without cells for scenes of a handful of primitives (build_grid() declines below 32 griddable ones): EVERYTHING
// sits in the always-list, the box is a point no ray is asked to walk through.  It exists for the end of a list-scan launch:
// the parked paths of such a scene are then finished lane per ray by the resume pass (4 exact tests per segment on test1.txt)
// instead of by the tail kernel with its 8 lanes per ray (0.55 of the 1.55 ms of BASELINE configuration 2).  Exact by the same
// argument as the always-list of any grid.  false: too many primitives for an always-list.
template <typename F>
inline bool build_degenerate_grid(int n_sph, int n_sph_pad, int n_msph, int n_tri, std::vector<uint32_t> &cell_start, std::vector<GridPrim> &cell_prims, std::vector<uint32_t> &always, GridRec<F> &G)
{
    if ((int64_t)n_sph + n_msph + n_tri > 48) return false;
    always.clear();
    for (int i = 0; i < n_sph; ++i) always.push_back((uint32_t)i);
    for (int i = 0; i < n_msph; ++i) always.push_back((uint32_t)(n_sph_pad + i));
    for (int i = 0; i < n_tri; ++i) always.push_back((uint32_t)(n_sph_pad + n_msph + i));
    cell_start.assign(2, 0u);
    cell_prims.assign(1, (GridPrim)0);
    G = GridRec<F>();
    for (int k = 0; k < 3; ++k) G.gmin[k] = G.gmax[k] = G.center[k] = (F)0, G.cell[k] = G.inv_cell[k] = (F)1, G.dims[k] = 1;
    G.far2 = std::numeric_limits<F>::infinity(); // no ray is "far": there is nothing in the cells it could miss
    G.slack = G.slack1 = G.half_diag = (F)0;
    G.max_steps = 4 + (int)always.size();
    G.dir2_max = (F)(sizeof(F) == 4 ? 1e15 : 1e120);
    G.walk_slice = 4;
    return true;
}

} // namespace rrtx

#endif
