// Packing of the reference-layout scene tables (include/rrtx.h) into the device records of
// rrtx_device.h: what create_world builds from the same tables (rrt.cu:124-174), minus the heap.
// Header-only: rrtx_api.cpp uploads the result; tests/path_host_render.cpp feeds it to the path
// arithmetic compiled for the host.
#ifndef RRTX_PACK_H
#define RRTX_PACK_H

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/rrtx.h"
#include "rrtx_device.h"

namespace rrtx {

template <typename F> struct RefTypes;
template <> struct RefTypes<float> {
    typedef rrtx_camera_f32 camera;
    typedef rrtx_material_f32 material;
    typedef rrtx_sphere_f32 sphere;
    typedef rrtx_moving_sphere_f32 moving_sphere;
    typedef rrtx_triangle_f32 triangle;
};
template <> struct RefTypes<double> {
    typedef rrtx_camera_f64 camera;
    typedef rrtx_material_f64 material;
    typedef rrtx_sphere_f64 sphere;
    typedef rrtx_moving_sphere_f64 moving_sphere;
    typedef rrtx_triangle_f64 triangle;
};


template <typename F> void pack_unit(const F v[3], F out[3])
{
    // vec3.h:125 unit_vector = (1/len) * v
    F len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    F inv = (F)1 / len;
    out[0] = inv * v[0], out[1] = inv * v[1], out[2] = inv * v[2];
}

// ---------------------------------------------------------------------------------------------
// The scan filter on the matrix cores (render_kernel, LDSMODE = 3).  The filter's value
//     f = (c.n)^2 + b.c + g - thr        (candidate <=> not f < 0; rrtx_path.h: make_filter_ray, filter_value)
// is ONE dot product once (c.n)^2 is written out in the six monomials:  sum_{i<=j} m_ij c_i c_j . n_i n_j  (m = 1, 2), and
// the matrix cores evaluate 32 spheres x 32 rays of it at once (two chained v_mfma_f32_32x32x16_f16) if every f32 operand x is split into two f16 pieces,
// x = x_h + x_l (22 bits), and the three leading cross products x_h y_h + x_h y_l + x_l y_h are kept: 6 x 3 + 3 x 3 terms, two for
// g (x 1), two for thr (x -1): 31 of the instruction's 32.  What is dropped (x_l y_l: 2^-22 of a term) and how the products are
// added up (f32) is far inside the margins the filter has anyway (K eps (|o|^2 + |c|^2 + r^2), K = kFilterKMf: the numerical search of
// tests/test_filter_mfma.py finds no false negative from K = 16 up).  Spheres the f16 operands cannot hold (a monomial or the
// threshold beyond 60000: the r = 1000 ground sphere) or resolve (r^2 < 1e-3) are listed apart (`big`) and tested exactly.
//
// Table: 32 halves per sphere, [block of 32 spheres][half of the terms (16)][lane][8]: lane l of a wave reads 16 bytes at
// ((block x 2 + half) x 64 + l) x 16 - sphere l % 32, terms 16 half + 8 (l / 32) ... + 7 - the A operand of v_mfma_f32_32x32x16_f16 as it
// wants it; two of them, chained, cover the 32 terms.  Padded to whole blocks with records nothing is a candidate for.
// Term order (sphere half, ray half):  q in xx yy zz xy xz yz: (Q_h, N_h) (Q_h, N_l) (Q_l, N_h);  i in x y z: (c_h, b_h) (c_h, b_l) (c_l, b_h);
// (1, g_h) (1, g_l);  (thr_h, -1) (thr_l, -1);  (0, 0).
// ---------------------------------------------------------------------------------------------
// The margins' budget (DESIGN.md 3a has the derivation, tests/test_filter_derivation.py holds every line against measurement), in units of
// u S, u = 2^-24 (the unit roundoff of fp32), S = |o|^2 + |c|^2 + r^2.  A filter is safe iff its margin K u S covers the rounding error of
// the reference's discriminant (sphere.h:35-41, per |d|^2) plus that of the filter's own value:
constexpr int kBudgetRef = 42;               // the reference's discriminant in fp32: u |d|^2 (21 |o - c|^2 + 6 r^2), |o - c|^2 <= 2 (|o|^2 + |c|^2)   (fp64 rays: 2^-29 of that)
constexpr int kBudgetIn64 = 16;              // fp64 rays: origin, direction and centres rounded to float before the fp32 filter sees them
constexpr int kBudgetRay = 33;               // the ray's side in fp32: n = d / |d| (4.5 u a component), s = o.n (7.5 u |o|), b, g, the products n_i n_j
constexpr int kBudgetVec = 11;               // vector form: c.n, the three FMAs over b.c + g, the last FMA
constexpr int kBudgetMfSplit = 36;           // matrix form: operands as two f16 pieces (2^-22 each way), the low x low products dropped
constexpr int kBudgetMfSum = 93;             // matrix form: 31 exact products added in f32, 30 roundings of a partial sum <= 3.003 lambda S, in any order
constexpr int kBudgetMfUnderflowRay = 2;     // matrix form: f16 underflow (2^-25 absolute per piece) of the ray's pieces: 2 u S and 2e-7 absolute
constexpr int kBudgetMfUnderflowSphere = 150; // matrix form: the same for a sphere's pieces - set aside; mf_sphere_stays_in_table() holds a sphere to it
static_assert(kBudgetRef + kBudgetRay + kBudgetVec <= kFilterK, "vector form, fp32 rays");
static_assert(kBudgetIn64 + kBudgetRay + kBudgetVec <= kFilterK64, "vector form, fp64 rays");
static_assert(kBudgetRef + kBudgetRay + kBudgetMfSplit + kBudgetMfSum + kBudgetMfUnderflowRay + kBudgetMfUnderflowSphere <= kFilterKMf, "matrix form, fp32 rays");
static_assert(kBudgetIn64 + kBudgetRay + kBudgetMfSplit + kBudgetMfSum + kBudgetMfUnderflowRay + kBudgetMfUnderflowSphere <= kFilterKMf64, "matrix form, fp64 rays");
// A sphere's f16 pieces lose up to 2^-25 each to underflow, and that loss meets the ray's operands, which the kernel scales to under 2^14: per
// unit of the scale lambda the error is 2^-25 U max(1, 2 m / 2^14), U the sum of the sphere's ten operands' magnitudes, m <= 1.0001 |o|^2 the
// largest of the ray's.  It fits the share set aside for it - 150 u (|o|^2 + |c|^2 + r^2) and 9.8e-6 of the threshold's absolute term 1e-5 - iff
//     2^-25 U <= 9.8e-6 + 150 u (|c|^2 + r^2)      (rays the kernel does not scale)       and       U <= 2.4e6      (rays it does).
// A sphere that fails either is listed apart and tested exactly.  (Under the caps below - every operand within 60 000 - none does: the rule says so, not the caps.)
inline bool mf_sphere_stays_in_table(long double operand_sum, long double c2, long double r2)
{
    return 0x1p-25L * operand_sum <= 9.8e-6L + (long double)kBudgetMfUnderflowSphere * 0x1p-24L * (c2 + r2) && operand_sum <= 2.4e6L;
}

inline uint16_t f32_to_f16_bits(float f) // round to nearest even; overflow -> inf; subnormals kept
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return (uint16_t)(sign | (mag > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (mag >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); // >= 65520: rounds to inf
    if (mag < 0x33000001u) return (uint16_t)sign;             // < 2^-25 (or == 2^-25: ties to even = 0)
    int e = (int)(mag >> 23) - 127;
    uint32_t m = (mag & 0x7FFFFFu) | 0x800000u; // 24 bits
    int shift = e >= -14 ? 13 : 13 + (-14 - e); // bits to drop
    uint32_t half = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) half += 1;
    if (e >= -14) return (uint16_t)(sign | (uint32_t)(((e + 15) << 10) + (half - 0x400u))); // (a carry out of the mantissa lands in the exponent)
    return (uint16_t)(sign | half); // subnormal (a carry makes it the smallest normal)
}
inline float f16_bits_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    float out;
    if (e == 0) {
        out = (float)m * 0x1p-24f;
        if (sign) out = -out;
        return out;
    }
    const uint32_t x = e == 31 ? (sign | 0x7F800000u | (m << 13)) : (sign | ((e + 112u) << 23) | (m << 13));
    memcpy(&out, &x, 4);
    return out;
}
struct MfTable {
    std::vector<uint16_t> halves; // mf_padded(n_pad) x 32, in the layout above
    std::vector<uint32_t> big;    // spheres the table cannot hold: tested exactly by every segment
    bool ok = false;
};
constexpr int kMfBlock = 32; // spheres per block of the table
inline int mf_padded(int n_pad) { return (n_pad + kMfBlock - 1) / kMfBlock * kMfBlock; }
template <typename F> inline void pack_mf_table(const std::vector<SphereHot<F>> &hot, int n_sph, int n_pad_scan, MfTable &out)
{
    const int n_pad = mf_padded(n_pad_scan > 0 ? n_pad_scan : 1);
    out.ok = false, out.big.clear();
    out.halves.assign((size_t)n_pad * 32, 0);
    const long double eps = 0x1p-24L, K = sizeof(F) == 4 ? (long double)kFilterKMf : (long double)kFilterKMf64;
    auto put = [&](int i, int term, uint16_t v) { out.halves[((((size_t)(i / 32) * 2 + (size_t)(term / 16)) * 64 + (size_t)(i % 32) + 32 * (size_t)((term % 16) / 8)) * 8) + (size_t)(term % 8)] = v; };
    auto split = [&](long double x, uint16_t &h, uint16_t &l) {
        h = f32_to_f16_bits((float)x);
        l = f32_to_f16_bits((float)(x - (long double)f16_bits_to_f32(h)));
    };
    for (int i = 0; i < n_pad; ++i) {
        bool in_table = i < n_sph;
        long double v[10] = {0}, thr = 0;
        if (in_table) {
            const long double cx = hot[i].cx, cy = hot[i].cy, cz = hot[i].cz, r2 = hot[i].r2;
            const long double c2 = cx * cx + cy * cy + cz * cz;
            v[0] = cx * cx, v[1] = cy * cy, v[2] = cz * cz, v[3] = 2 * cx * cy, v[4] = 2 * cx * cz, v[5] = 2 * cy * cz, v[6] = cx, v[7] = cy, v[8] = cz;
            // (the absolute term: what the f16 pieces lose to underflow near zero)
            thr = (c2 - r2) - K * eps * (c2 + r2) - 1e-5L;
            for (int k = 0; k < 9; ++k) in_table = in_table && std::isfinite((double)v[k]) && std::fabs((double)v[k]) <= 60000.0;
            in_table = in_table && std::isfinite((double)thr) && std::fabs((double)thr) <= 60000.0 && (double)r2 >= 1e-3;
            long double operand_sum = std::fabs((double)thr);
            for (int k = 0; k < 9; ++k) operand_sum += std::fabs((double)v[k]);
            in_table = in_table && mf_sphere_stays_in_table(operand_sum, c2, r2); // (the margin's derivation: see above)
            if (!in_table) out.big.push_back((uint32_t)i);
        }
        if (!in_table) { // a padding record, or a sphere listed apart: f = -60000 whatever the ray, never a candidate
            put(i, 29, f32_to_f16_bits(60000.0f));
            continue;
        }
        uint16_t h, l;
        for (int q = 0; q < 9; ++q) {
            split(v[q], h, l);
            put(i, 3 * q + 0, h), put(i, 3 * q + 1, h), put(i, 3 * q + 2, l);
        }
        put(i, 27, f32_to_f16_bits(1.0f)), put(i, 28, f32_to_f16_bits(1.0f));
        split(thr, h, l);
        // (the pieces must not add up to more than thr: the lower piece is rounded to nearest - one f16 ulp of it, 2^-22 of thr, is inside the margin)
        put(i, 29, h), put(i, 30, l);
    }
    out.ok = out.big.size() <= 16;
}

template <typename F> struct PackedScene {
    std::vector<MaterialRec<F>> mat;
    std::vector<SphereHot<F>> hot;        // exact-test records {c, r*r}
    std::vector<SphereHot<float>> filter; // conservative-filter records {c, thr}: fp32 for every F
    std::vector<SphereCold<F>> cold;
    std::vector<MovingSphereRec<F>> ms;
    std::vector<TriangleRec<F>> tri;
    std::vector<TriScanRec<F>> tri_scan;
    CameraRec<F> cam;
    int n_pad = 0;          // spheres padded to a multiple of kSpherePad with never-hit records
    bool filter_ok = true;  // magnitudes within the filter's proven range
    bool tail_ok = true;    // finite, bounded primitives: no NaN roots possible for sane rays
};

// Returns nullptr, or what is wrong with the tables.
template <typename F> const char *pack_scene(const rrtx_scene_desc *s, PackedScene<F> &out)
{
    typedef RefTypes<F> R;
    const typename R::camera *cam = (const typename R::camera *)s->camera;
    const typename R::material *mats = (const typename R::material *)s->materials;
    const typename R::sphere *sph = (const typename R::sphere *)s->spheres;
    const typename R::moving_sphere *msp = (const typename R::moving_sphere *)s->moving_spheres;
    const typename R::triangle *tri = (const typename R::triangle *)s->triangles;

    // camera.h:43-48 has the same member order as CameraRec
    static_assert(sizeof(typename R::camera) == sizeof(CameraRec<F>), "camera layout");
    memcpy(&out.cam, cam, sizeof(CameraRec<F>));

    // materials: what create_world builds (rrt.cu:137-148; material.h:19,48,74)
    std::vector<MaterialRec<F>> &hmat = out.mat;
    hmat.clear(), hmat.resize(s->num_materials > 0 ? s->num_materials : 1);
    for (int i = 0; i < s->num_materials; ++i) {
        MaterialRec<F> m = {};
        m.type = mats[i].type;
        if (mats[i].type == RRTX_LAMBERTIAN) {
            m.r = mats[i].mat.lambertian.albedo[0], m.g = mats[i].mat.lambertian.albedo[1], m.b = mats[i].mat.lambertian.albedo[2];
        }
        else if (mats[i].type == RRTX_METAL) {
            m.r = mats[i].mat.metal.albedo[0], m.g = mats[i].mat.metal.albedo[1], m.b = mats[i].mat.metal.albedo[2];
            F f = (F)mats[i].mat.metal.fuzz;
            m.param = f < (F)1.0 ? f : (F)1.0;
        }
        else if (mats[i].type == RRTX_DIELECTRIC) {
            m.param = (F)mats[i].mat.dielectric.ref_idx;
        }
        else
            return "rrtx_set_scene: unknown material type";
        hmat[i] = m;
    }

    auto check_mat = [&](int idx) { return idx >= 0 && idx < s->num_materials; };

    // spheres: hot {center, r*r} + cold {r, material}; padded with never-hit records
    const int pad = kSpherePad;
    const int n_pad = ((s->num_spheres + pad - 1) / pad) * pad;
    std::vector<SphereHot<F>> &hhot = out.hot;
    hhot.clear(), hhot.resize(n_pad > 0 ? n_pad : 1);
    std::vector<SphereCold<F>> &hcold = out.cold;
    hcold.clear(), hcold.resize(n_pad > 0 ? n_pad : 1);
    for (int i = 0; i < n_pad; ++i) {
        if (i < s->num_spheres) {
            if (!check_mat(sph[i].material_idx)) return "rrtx_set_scene: sphere material index out of range";
            F r = (F)sph[i].radius; // sphere(cen, FP_T r, m), sphere.h:11
            hhot[i].cx = sph[i].center[0], hhot[i].cy = sph[i].center[1], hhot[i].cz = sph[i].center[2];
            hhot[i].r2 = r * r; // sphere.h:38
            hcold[i].radius = r;
            hcold[i].mat = sph[i].material_idx;
        }
        else {
            // c = |oc|^2 + inf => discriminant = -inf: never a candidate (and phase 2 skips k >= n_sph)
            hhot[i].cx = hhot[i].cy = hhot[i].cz = 0;
            hhot[i].r2 = -std::numeric_limits<F>::infinity();
            hcold[i].radius = 1;
            hcold[i].mat = 0;
        }
    }
    // Conservative scan filter table {c, thr} — in float for every F: thr = |c|^2 - r^2 - K eps32 (|c|^2 + r^2),
    // evaluated in long double from the F-precision centre and radius and rounded DOWN, so the device-side
    // test can only err towards "candidate" (K = kFilterK for fp32 rays; kFilterK64 for fp64 rays, whose
    // components the filter rounds to float).  The bound assumes finite, not absurdly scaled magnitudes;
    // scenes outside that range use the exact scan.
    std::vector<SphereHot<float>> &hfil = out.filter;
    hfil.clear(), hfil.resize(n_pad > 0 ? n_pad : 1);
    bool &filter_ok = out.filter_ok;
    filter_ok = true;
    {
        const long double eps = 0x1p-24L, K = sizeof(F) == 4 ? (long double)kFilterK : (long double)kFilterK64;
        const long double big = 1e30L, tiny = 1e-25L;
        for (int i = 0; i < n_pad; ++i) {
            hfil[i].cx = (float)hhot[i].cx, hfil[i].cy = (float)hhot[i].cy, hfil[i].cz = (float)hhot[i].cz;
            if (i >= s->num_spheres) {
                hfil[i].r2 = std::numeric_limits<float>::infinity(); // finite test values are always below it
                continue;
            }
            const long double cx = hhot[i].cx, cy = hhot[i].cy, cz = hhot[i].cz, r2 = hhot[i].r2;
            const long double c2 = cx * cx + cy * cy + cz * cz;
            if (!(std::isfinite((double)c2) && std::isfinite((double)r2)) || !(c2 + r2 <= big) || !(c2 + r2 >= tiny)) filter_ok = false;
            const long double thr = (c2 - r2) - K * eps * (c2 + r2);
            float t = (float)thr;
            if ((long double)t > thr) t = std::nextafter(t, -std::numeric_limits<float>::infinity());
            t = std::nextafter(t, -std::numeric_limits<float>::infinity()); // one more ulp of slack
            hfil[i].r2 = t;
        }
    }
    std::vector<MovingSphereRec<F>> &hms = out.ms;
    hms.clear(), hms.resize(s->num_moving_spheres > 0 ? s->num_moving_spheres : 1);
    for (int i = 0; i < s->num_moving_spheres; ++i) {
        if (!check_mat(msp[i].material_idx)) return "rrtx_set_scene: moving sphere material index out of range";
        MovingSphereRec<F> m = {};
        F t0 = (F)msp[i].time0, t1 = (F)msp[i].time1, r = (F)msp[i].radius; // moving_sphere.h:11-12
        for (int k = 0; k < 3; ++k) {
            m.c0[k] = msp[i].center0[k];
            m.dc[k] = msp[i].center1[k] - msp[i].center0[k];
        }
        m.t0 = t0;
        m.dt = t1 - t0;
        m.r2 = r * r;
        m.radius = r;
        m.mat = msp[i].material_idx;
        hms[i] = m;
    }
    std::vector<TriangleRec<F>> &htri = out.tri;
    htri.clear(), htri.resize(s->num_triangles > 0 ? s->num_triangles : 1);
    for (int i = 0; i < s->num_triangles; ++i) {
        if (!check_mat(tri[i].material_idx)) return "rrtx_set_scene: triangle material index out of range";
        TriangleRec<F> t = {};
        F e1[3], e2[3], u1[3], u2[3], cr[3];
        for (int k = 0; k < 3; ++k) {
            t.v0[k] = tri[i].vertices[0][k];
            e1[k] = tri[i].vertices[1][k] - tri[i].vertices[0][k];
            e2[k] = tri[i].vertices[2][k] - tri[i].vertices[0][k];
            t.e1[k] = e1[k];
            t.e2[k] = e2[k];
        }
        // triangle.h:9-15: unit(cross(unit(v1-v0), unit(v2-v0)))
        pack_unit<F>(e1, u1);
        pack_unit<F>(e2, u2);
        cr[0] = u1[1] * u2[2] - u1[2] * u2[1];
        cr[1] = u1[2] * u2[0] - u1[0] * u2[2];
        cr[2] = u1[0] * u2[1] - u1[1] * u2[0];
        pack_unit<F>(cr, t.n);
        t.mat = tri[i].material_idx;
        htri[i] = t;
    }
    out.tri_scan.clear(), out.tri_scan.resize(htri.size());
    for (size_t i = 0; i < htri.size(); ++i) {
        TriScanRec<F> r = {};
        for (int k = 0; k < 3; ++k) r.v0[k] = htri[i].v0[k], r.e1[k] = htri[i].e1[k], r.e2[k] = htri[i].e2[k];
        out.tri_scan[i] = r;
    }

    out.n_pad = n_pad;
    {
        // the tail kernel's split scan needs every root to be non-NaN: finite primitives of bounded
        // magnitude (so that no intermediate of the discriminant overflows)
        const double lim = sizeof(F) == 4 ? 3e7 : 1e60;
        bool ok = true;
        auto chk = [&](double v) { ok = ok && std::isfinite(v) && std::fabs(v) <= lim; };
        for (int i = 0; i < s->num_spheres; ++i) chk(hhot[i].cx), chk(hhot[i].cy), chk(hhot[i].cz), chk(hcold[i].radius);
        for (int i = 0; i < s->num_moving_spheres; ++i) {
            for (int k = 0; k < 3; ++k) chk(hms[i].c0[k]), chk(hms[i].dc[k]);
            chk(hms[i].t0), chk(hms[i].dt), chk(hms[i].radius);
            ok = ok && hms[i].dt != 0; // (time - t0) / 0 is where NaN centres come from
        }
        for (int i = 0; i < s->num_triangles; ++i)
            for (int k = 0; k < 3; ++k) chk(htri[i].v0[k]), chk(htri[i].e1[k]), chk(htri[i].e2[k]), chk(htri[i].n[k]);
        out.tail_ok = ok;
    }
    return nullptr;
}

} // namespace rrtx

#endif
