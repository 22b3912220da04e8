// =====================================================================================
// ORACLE SIDE — TEST INFRASTRUCTURE ONLY.  The reference-side binding of INTEGRATION.md §1, as
// compiled code: class Rrt of the reference's rrt.h:14-48 implemented on top of librrtx.so.
//
// This is the file a maintainer of the reference adds IN PLACE OF rrt.cu / rrt.cpp.  oracle/Makefile
// builds it together with the reference's own, unchanged main.cpp (argv parsing, scene.h parser,
// color.h quantiser, stb PNG writer - compiled where they lie under /root/reference, never copied)
// into oracle/_ref/rrt_dropin (fp32, -DUSE_FLOAT) and oracle/_ref/rrtd_dropin (fp64):
//
//     g++ -O3 -DUSE_FLOAT -I/root/reference /root/reference/main.cpp oracle/ref_dropin.cpp -lrrtx
//
// so that the seam is exercised by the reference's own `main`: scene::scene parses, the five POD
// tables of rrt.cu:217-247 go through rrtx_set_scene() as they are (no conversion: the structs of
// include/rrtx.h are layout-identical, checked below at compile time), rrtx_render() fills the
// `vec3 fb[w*h]` that main.cpp:140-167 quantises and writes.  tests/test_gpu_dropin.py requires the
// PPM bytes / PNG pixels of this binary to equal those of the product's own `rrt` / `rrtd`.
//
// Built WITHOUT -DUSE_CUDA (that branch of rrt.h / main.cpp needs the CUDA headers, which this image
// does not have): the constructor is the 5-argument one of rrt.h:16-31, -tx / -ty / -q / -D do not
// exist in this `main`; with USE_CUDA the two extra members are forwarded below all the same.
// =====================================================================================
#include "rrt.h" // the reference's (via -I): class Rrt, scene, camera, vec3

#include "../include/rrtx.h"

#include <cstdlib>
#include <iostream>
#include <vector>

// the layouts rrtx.h promises (include/rrtx.h:40-97) against the reference's own types
#ifdef USE_FLOAT
typedef rrtx_camera_f32 x_camera;
typedef rrtx_material_f32 x_material;
typedef rrtx_sphere_f32 x_sphere;
typedef rrtx_moving_sphere_f32 x_msphere;
typedef rrtx_triangle_f32 x_triangle;
#else
typedef rrtx_camera_f64 x_camera;
typedef rrtx_material_f64 x_material;
typedef rrtx_sphere_f64 x_sphere;
typedef rrtx_moving_sphere_f64 x_msphere;
typedef rrtx_triangle_f64 x_triangle;
#endif
static_assert(sizeof(camera) == sizeof(x_camera), "camera.h:43-48 vs rrtx_camera");
static_assert(sizeof(scene_material) == sizeof(x_material), "scene.h:183-208 vs rrtx_material");
static_assert(sizeof(scene_sphere) == sizeof(x_sphere), "scene.h:43-47 vs rrtx_sphere");
static_assert(sizeof(scene_moving_sphere) == sizeof(x_msphere), "scene.h:49-54 vs rrtx_moving_sphere");
static_assert(sizeof(scene_instance_triangle) == sizeof(x_triangle), "scene.h:27-41 vs rrtx_triangle");
static_assert(sizeof(vec3) == 3 * sizeof(FP_T), "vec3 is the framebuffer element rrtx_render writes");

static void check(int rc) // check_cuda, rrt.cu:31-40
{
    if (rc) {
        std::cerr << "HIP error = " << rc << " : " << rrtx_last_error() << "\n";
        std::exit(99);
    }
}

vec3 *Rrt::render(scene *the_scene) // rrt.cu:186 / rrt.cpp:99
{
    // the five tables rrt.cu:217-247 copies into managed memory, as plain host arrays
    std::vector<scene_material> mats;
    for (auto m : the_scene->materials) mats.push_back(*m);
    std::vector<scene_sphere> sph;
    for (auto s : the_scene->spheres) sph.push_back(*s);
    std::vector<scene_moving_sphere> ms;
    for (auto s : the_scene->moving_spheres) ms.push_back(*s);
    std::vector<scene_instance_triangle> tri(the_scene->num_triangles() + 1);
    the_scene->fill_instance_triangles(tri.data());

    rrtx_scene_desc d = {};
    d.fp64 = sizeof(FP_T) == 8;
    d.camera = the_scene->cam; // camera.h:43-48 == rrtx_camera_f32 / _f64
    d.num_materials = (int)mats.size(), d.materials = mats.data();
    d.num_spheres = (int)sph.size(), d.spheres = sph.data();
    d.num_moving_spheres = (int)ms.size(), d.moving_spheres = ms.data();
    d.num_triangles = the_scene->num_triangles(), d.triangles = tri.data();

    rrtx_params p = {};
    p.image_width = image_width, p.image_height = image_height;
    p.samples_per_pixel = samples_per_pixel, p.max_depth = max_depth, p.use_bvh = bvh;
#ifdef USE_CUDA
    p.threads_x = num_threads_x, p.threads_y = num_threads_y;
#else
    p.threads_x = p.threads_y = 8;
#endif
    p.fp64 = d.fp64, p.collect_stats = 1;

    std::cerr << "Rendering a " << image_width << "x" << image_height << " image with " << samples_per_pixel << " samples per pixel through librrtx ("
              << rrtx_version() << ").\n"; // rrt.cu:198-202
    rrtx_ctx *ctx = nullptr;
    check(rrtx_abi_version() == RRTX_ABI_VERSION ? 0 : RRTX_E_UNSUPPORTED); // the library writes sizeof(ITS rrtx_stats) below
    check(rrtx_create(&p, &ctx));                      // Rrt::Rrt
    check(rrtx_set_scene(ctx, &d));                    // replaces create_world<<<1,1>>>, rrt.cu:266
    fb = new vec3[(size_t)image_width * image_height]; // vec3 = 3 x FP_T: the layout rrtx_render writes
    rrtx_stats st;
    check(rrtx_render(ctx, fb, &st)); // render_init + cuda_render, rrt.cu:291-298
    std::cerr << "took " << st.kernel_ms / 1000.0 << " seconds.\n";
    rrtx_destroy(ctx); // free_world + cudaFree, rrt.cu:324-331
    return fb;         // row 0 = bottom, un-normalised sums (rrt.cu:118)
}

Rrt::~Rrt() { delete[] fb; }
