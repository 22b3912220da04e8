// =====================================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Thin C entry points around the reference's OWN sources, compiled where they lie under
// /root/reference (never copied): this translation unit textually includes the reference's
// rrt.cpp (which pulls in scene.h, camera.h, sphere.h, material.h, ...) and color.h through
// the include path, with oracle/ref_hook.h force-included in front so that random_uniform()
// (rtweekend.h:64-69) draws from the product's counter-based stream.  The functions below only
// drive reference code:
//   * ref_scene_*   -> scene::scene (scene.h:212), fill_instance_triangles (scene.h:467)
//   * ref_render    -> create_world (rrt.cpp:54), camera::get_ray (camera.h:31),
//                      ray_color (rrt.cpp:25); the 6-line sample loop of rrt.cpp:139-148 is the
//                      only thing re-stated here, because the stream has to be re-keyed per
//                      (pixel, sample)
//   * ref_quantise / ref_ppm -> convert_color / write_color (color.h:8-32), main.cpp:140-162 order
// Output goes to oracle/_ref/ only (git-ignored).  Built with ROCm clang++ so that
// vec3(random(), random(), random()) draws x, y, z in source order.
// =====================================================================================
#include "rrt.cpp"

#include "color.h"

#include <cstring>

extern "C" {

void *ref_scene_load(const char *path, int w, int h) { return new scene(path, w, h); }

void ref_scene_free(void *s) { delete (scene *)s; }

void ref_scene_counts(void *sv, int counts[6])
{
    scene *s = (scene *)sv;
    counts[0] = (int)s->materials.size();
    counts[1] = (int)s->spheres.size();
    counts[2] = (int)s->moving_spheres.size();
    counts[3] = s->num_triangles();
    counts[4] = (int)s->objs.size();
    counts[5] = (int)s->obj_insts.size();
}

int ref_sizeof(int what)
{
    switch (what) {
    case 0: return (int)sizeof(FP_T);
    case 1: return (int)sizeof(camera);
    case 2: return (int)sizeof(scene_material);
    case 3: return (int)sizeof(scene_sphere);
    case 4: return (int)sizeof(scene_moving_sphere);
    case 5: return (int)sizeof(scene_instance_triangle);
    }
    return -1;
}

// raw bytes of the reference's POD tables, exactly what rrt.cu:217-247 marshals
void ref_scene_raw(void *sv, void *cam, void *mats, void *sph, void *msph, void *tris)
{
    scene *s = (scene *)sv;
    if (cam) memcpy(cam, (const void *)s->cam, sizeof(camera));
    if (mats)
        for (size_t i = 0; i < s->materials.size(); ++i)
            memcpy((char *)mats + i * sizeof(scene_material), (const void *)s->materials[i], sizeof(scene_material));
    if (sph)
        for (size_t i = 0; i < s->spheres.size(); ++i)
            memcpy((char *)sph + i * sizeof(scene_sphere), (const void *)s->spheres[i], sizeof(scene_sphere));
    if (msph)
        for (size_t i = 0; i < s->moving_spheres.size(); ++i)
            memcpy((char *)msph + i * sizeof(scene_moving_sphere), (const void *)s->moving_spheres[i], sizeof(scene_moving_sphere));
    if (tris) {
        std::vector<scene_instance_triangle> t(s->num_triangles() + 1);
        s->fill_instance_triangles(t.data());
        memcpy(tris, (const void *)t.data(), (size_t)s->num_triangles() * sizeof(scene_instance_triangle));
    }
}

// same table layout as rrto_scene_dump (values widened to double)
void ref_scene_dump(void *sv, double *cam24, double *mats, double *sph, double *msph, double *tris)
{
    scene *s = (scene *)sv;
    if (cam24) {
        const FP_T *c = (const FP_T *)(const void *)s->cam; // camera.h:43-48: 7 vec3 + 3 scalars
        for (int k = 0; k < 24; ++k) cam24[k] = c[k];
    }
    if (mats)
        for (size_t i = 0; i < s->materials.size(); ++i) {
            scene_material *m = s->materials[i];
            double *o = mats + 6 * i;
            o[0] = (int)m->type;
            o[1] = o[2] = o[3] = o[4] = o[5] = 0;
            if (m->type == LAMBERTIAN) {
                o[1] = m->mat.lambertian.albedo.x(), o[2] = m->mat.lambertian.albedo.y(), o[3] = m->mat.lambertian.albedo.z();
            }
            else if (m->type == METAL) {
                o[1] = m->mat.metal.albedo.x(), o[2] = m->mat.metal.albedo.y(), o[3] = m->mat.metal.albedo.z();
                o[4] = m->mat.metal.fuzz;
            }
            else
                o[5] = m->mat.dielectric.ref_idx;
        }
    if (sph)
        for (size_t i = 0; i < s->spheres.size(); ++i) {
            scene_sphere *p = s->spheres[i];
            double *o = sph + 5 * i;
            o[0] = p->center.x(), o[1] = p->center.y(), o[2] = p->center.z(), o[3] = p->radius, o[4] = p->material_idx;
        }
    if (msph)
        for (size_t i = 0; i < s->moving_spheres.size(); ++i) {
            scene_moving_sphere *p = s->moving_spheres[i];
            double *o = msph + 10 * i;
            o[0] = p->center0.x(), o[1] = p->center0.y(), o[2] = p->center0.z();
            o[3] = p->center1.x(), o[4] = p->center1.y(), o[5] = p->center1.z();
            o[6] = p->time0, o[7] = p->time1, o[8] = p->radius, o[9] = p->material_idx;
        }
    if (tris) {
        int n = s->num_triangles();
        std::vector<scene_instance_triangle> t(n + 1);
        s->fill_instance_triangles(t.data());
        for (int i = 0; i < n; ++i) {
            double *o = tris + 10 * i;
            for (int c = 0; c < 3; ++c) {
                o[3 * c + 0] = t[i].vertices[c].x(), o[3 * c + 1] = t[i].vertices[c].y(), o[3 * c + 2] = t[i].vertices[c].z();
            }
            o[9] = t[i].material_idx;
        }
    }
}

// fb: W*H*3 FP_T, row 0 = bottom (rrt.cpp:148); rows [row0,row1) are rendered
int ref_render(void *sv, int W, int H, int spp, int max_depth, uint32_t seed, int use_bvh, int row0, int row1, FP_T *fb)
{
    scene *s = (scene *)sv;
    hittable *world = create_world(s, use_bvh != 0); // rrt.cpp:54
    camera cam(*(s->cam));                           // rrt.cpp:112
    for (int j = row0; j < row1; ++j) {
        for (int i = 0; i < W; ++i) {
            color pixel_color(0, 0, 0);
            for (int sm = 0; sm < spp; ++sm) {
                rrtx_hook::open(seed, (uint32_t)(j * W + i), (uint32_t)sm);
                auto u = (i + random_uniform()) / (W - 1); // rrt.cpp:140
                auto v = (j + random_uniform()) / (H - 1); // rrt.cpp:141
                ray r = cam.get_ray(u, v);                 // rrt.cpp:142
                pixel_color += ray_color(r, world, max_depth, false); // rrt.cpp:143
            }
            FP_T *o = fb + 3 * ((size_t)j * W + i);
            o[0] = pixel_color.x(), o[1] = pixel_color.y(), o[2] = pixel_color.z();
        }
    }
    return 0; // the world leaks exactly as in the reference (rrt.cpp never frees it)
}

// main.cpp:153-162
void ref_quantise(const FP_T *fb, int W, int H, int spp, uint8_t *rgb)
{
    for (int j = H - 1, k = 0; j >= 0; j--, k++)
        for (int i = 0; i < W; i++) {
            int red, grn, blu;
            convert_color(color(fb[3 * ((size_t)j * W + i)], fb[3 * ((size_t)j * W + i) + 1], fb[3 * ((size_t)j * W + i) + 2]), spp, &red, &grn,
                          &blu);
            uint8_t *o = rgb + 3 * ((size_t)k * W + i);
            o[0] = uint8_t(red), o[1] = uint8_t(grn), o[2] = uint8_t(blu);
        }
}

void ref_convert_color(const double rgb[3], int spp, int out[3])
{
    convert_color(color((FP_T)rgb[0], (FP_T)rgb[1], (FP_T)rgb[2]), spp, &out[0], &out[1], &out[2]);
}

// main.cpp:140-149 — PPM text into a file instead of stdout
int ref_ppm(const FP_T *fb, int W, int H, int spp, const char *path)
{
    std::ofstream out(path);
    if (!out.good()) return -1;
    out << "P3\n" << W << ' ' << H << "\n255\n";
    for (int j = H - 1; j >= 0; --j)
        for (int i = 0; i < W; ++i) {
            size_t p = 3 * ((size_t)j * W + i);
            write_color(out, color(fb[p], fb[p + 1], fb[p + 2]), spp);
        }
    return 0;
}

double ref_rng_probe(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n)
{
    rrtx_hook::open(seed, pixel, sample);
    FP_T v = 0;
    for (uint32_t k = 0; k <= n; ++k) v = random_uniform();
    return (double)v;
}

} // extern "C"
