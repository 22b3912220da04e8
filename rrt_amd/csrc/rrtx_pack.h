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
