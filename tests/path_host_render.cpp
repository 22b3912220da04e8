// The render path's arithmetic, compiled for the HOST from the headers the kernel is built from
// (rrt_amd/csrc/rrtx_path.h: RNG, camera ray, exact primitive tests, shading, task decoding, grid walk;
// rrtx_pack.h / rrtx_grid.h: table packing, grid builder), driven by a plain loop over pixels and
// samples.  TEST INFRASTRUCTURE: it lets the CPU test tier compare the product's own source with the
// oracle bit for bit (tests/test_host.py), which otherwise needs a GPU.  The product never runs this.
//
//   path_host_render <scene.txt> <w> <h> <spp> <depth> <fp64:0|1> <chunk> <mode:0 list|1 grid> <out.raw>
// writes w*h*3 values (float or double), row 0 = bottom, un-normalised sums — what rrtx_render returns.
// The scene is parsed by the product's host parser through the C ABI (link with -lrrtx).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/rrtx.h"
#include "../rrt_amd/csrc/rrtx_device.h"
#include "../rrt_amd/csrc/rrtx_grid.h"
#include "../rrt_amd/csrc/rrtx_pack.h"
#include "../rrt_amd/csrc/rrtx_path.h"

using namespace rrtx;

template <typename F> static int render(const rrtx_scene_desc &desc, int w, int h, int spp, int depth, int chunk, int mode, const char *out_path)
{
    PackedScene<F> ps;
    if (const char *what = pack_scene<F>(&desc, ps)) {
        std::fprintf(stderr, "pack_scene: %s\n", what);
        return 2;
    }
    KernelParams<F> P = {};
    P.sph_hot = ps.hot.data(), P.sph_filter = ps.filter.data(), P.sph_cold = ps.cold.data(), P.msph = ps.ms.data(), P.tri = ps.tri.data(), P.mat = ps.mat.data();
    P.n_sph = desc.num_spheres, P.n_sph_padded = ps.n_pad, P.n_msph = desc.num_moving_spheres, P.n_tri = desc.num_triangles;
    P.cam = ps.cam;
    P.W = w, P.H = h, P.spp = spp, P.max_depth = depth, P.seed = 1984;
    if (chunk <= 0 || chunk > spp) chunk = spp;
    P.chunk = chunk, P.chunks_per_pixel = (spp + chunk - 1) / chunk;
    P.local_rows = h, P.tile_rows = 4, P.shard_rank = 0, P.shard_count = 1;
    P.taper_pixel = (uint32_t)(w * h), P.taper_task_base = P.taper_pixel * (uint32_t)P.chunks_per_pixel, P.total_tasks = P.taper_task_base;
    P.div_cpp = make_fastdiv((uint32_t)P.chunks_per_pixel), P.div_spp = make_fastdiv((uint32_t)spp), P.div_w = make_fastdiv((uint32_t)w), P.div_tile = make_fastdiv(4u);
    std::vector<uint32_t> cell_start, always;
    std::vector<GridPrim> cell_prims;
    bool grid = false;
    if (mode == 1) {
        GridRec<F> G = {};
        grid = ps.tail_ok && build_grid<F>(ps.hot, ps.cold, desc.num_spheres, ps.n_pad, ps.ms, desc.num_moving_spheres, ps.tri, desc.num_triangles, ps.cam, cell_start, cell_prims, always, G);
        if (grid) {
            if (cell_prims.empty()) cell_prims.push_back(0);
            P.grid = G, P.grid_cell_start = cell_start.data(), P.grid_cell_prims = cell_prims.data(), P.grid_always = always.empty() ? nullptr : always.data();
            P.n_always = (int)always.size(), P.n_grid_cells = (int)P.grid.dims[0] * (int)P.grid.dims[1] * (int)P.grid.dims[2], P.n_grid_prims = (int)cell_start[(size_t)P.n_grid_cells];
        }
    }
    const F t_min = (F)0.001; // rrt.cpp:32 typing
    const int msph_base = P.n_sph_padded, tri_base = P.n_sph_padded + P.n_msph;
    std::vector<F> fb((size_t)w * h * 3, (F)0);
    unsigned long long segments = 0, walked = 0;
    for (uint32_t q = 0; q < (uint32_t)(w * h); ++q) {
        F sum[3] = {0, 0, 0};
        for (int c = 0; c < P.chunks_per_pixel; ++c) { // a task = (pixel, chunk); finalize_kernel adds the chunks in order
            const uint32_t task = q * (uint32_t)P.chunks_per_pixel + (uint32_t)c;
            int px_i, px_j, s_first, s_end;
            task_decode<F>(P, task, px_i, px_j, s_first, s_end);
            V3<F> acc = mk<F>(0, 0, 0);
            for (int s = s_first; s < s_end; ++s) {
                Rng rng = {0, 0, 0};
                Path<F> path;
                camera_ray<F>(P, px_i, px_j, s, rng, path);
                V3<F> radiance = mk<F>(0, 0, 0);
                bool done = depth <= 0; // rrt.cu:47: the loop body never runs
                while (!done) {
                    segments += 1;
                    const F a = vlen2<F>(path.d);
                    HitInfo<F> best = {std::numeric_limits<F>::infinity(), -1};
                    int r = kWalkNeedsScan;
                    if (grid) {
                        uint32_t cell = 0;
                        F t_out = 0;
                        bool resume = false;
                        do { // in slices of RRTX_WALK_SLICE cells, as the kernel walks
                            r = accel_closest_hit<F>(P, ps.hot.data(), cell_start.data(), cell_prims.data(), path, a, t_min, best, resume, cell, t_out, 4);
                            resume = true;
                        } while (r == kWalkGoesOn);
                        walked += r == kWalkDone;
                    }
                    if (r == kWalkNeedsScan || r == kWalkFarScan) { // hittable_list.h:95-117
                        best.t = std::numeric_limits<F>::infinity(), best.idx = -1;
                        for (int k = 0; k < P.n_sph; ++k) refine_sphere<F>(ps.hot[k].cx, ps.hot[k].cy, ps.hot[k].cz, ps.hot[k].r2, path, a, t_min, k, best);
                        for (int k = 0; k < P.n_msph; ++k) {
                            const V3<F> cen = msphere_center<F>(ps.ms[k], path.tm);
                            refine_sphere<F>(cen.x, cen.y, cen.z, ps.ms[k].r2, path, a, t_min, msph_base + k, best);
                        }
                        for (int k = 0; k < P.n_tri; ++k) {
                            F tt;
                            if (triangle_test<F, true>(ps.tri[k], path, t_min, best.t, tt)) best.t = tt, best.idx = tri_base + k;
                        }
                    }
                    done = shade<F>(P, best, path, rng, radiance);
                }
                acc = vadd<F>(acc, radiance); // rrt.cu:115
            }
            if (P.chunks_per_pixel == 1) // one task per pixel: its sum IS the pixel (no finalize pass; keeps a -0)
                sum[0] = acc.x, sum[1] = acc.y, sum[2] = acc.z;
            else
                sum[0] = sum[0] + acc.x, sum[1] = sum[1] + acc.y, sum[2] = sum[2] + acc.z;
        }
        F *o = &fb[(size_t)q * 3];
        o[0] = sum[0], o[1] = sum[1], o[2] = sum[2];
    }
    FILE *f = std::fopen(out_path, "wb");
    if (!f || std::fwrite(fb.data(), sizeof(F), fb.size(), f) != fb.size()) return 3;
    std::fclose(f);
    std::printf("segments %llu walked %llu grid %d\n", segments, walked, grid ? P.n_grid_cells : 0);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 10) {
        std::fprintf(stderr, "usage: path_host_render scene w h spp depth fp64 chunk mode out.raw\n");
        return 1;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), spp = std::atoi(argv[4]), depth = std::atoi(argv[5]), fp64 = std::atoi(argv[6]), chunk = std::atoi(argv[7]), mode = std::atoi(argv[8]);
    rrtx_scene *scene = nullptr;
    if (rrtx_scene_load(argv[1], w, h, fp64, &scene)) {
        std::fprintf(stderr, "cannot load %s: %s\n", argv[1], rrtx_last_error());
        return 1;
    }
    rrtx_scene_desc desc;
    rrtx_scene_describe(scene, &desc);
    const int rc = fp64 ? render<double>(desc, w, h, spp, depth, chunk, mode, argv[9]) : render<float>(desc, w, h, spp, depth, chunk, mode, argv[9]);
    rrtx_scene_free(scene);
    return rc;
}
