// Stand-in for the DEVICE half of include/rrtx.h, for ThreadSanitizer runs of the CLI's host logic only
// (tests/test_host.py: rrt_main.cpp + host_scene.cpp + host_image.cpp + this file, no GPU, no HIP).
// It renders nothing: a "frame" is a deterministic pattern of the scene's primitive counts, the camera and the
// pixel - enough to tell frames apart and to check that every image of a batch lands in its own file, in order.
// Calls take a few hundred microseconds of varying length so that parser, renderers and writers interleave.
// Never linked into the product; the product has no CPU path.
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/rrtx.h"

namespace {
thread_local std::string g_err;
std::atomic<int> g_calls{0};
void nap()
{
    const int k = g_calls.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 137 * (k % 11)));
}
struct Tables {
    int32_t n_sph = 0, n_msph = 0, n_tri = 0, n_mat = 0;
    double cam = 0;
};
template <typename F> void paint(const rrtx_params &p, const Tables &t, F *fb, int row0, int row1)
{
    for (int j = row0; j < row1; ++j)
        for (int i = 0; i < p.image_width; ++i) {
            F *px = fb + ((size_t)j * p.image_width + i) * 3;
            px[0] = (F)(((i * 7 + j * 13 + t.n_sph) % 256) / 255.0 * p.samples_per_pixel);
            px[1] = (F)(((i * 3 + j * 5 + t.n_tri + t.n_msph * 17) % 256) / 255.0 * p.samples_per_pixel);
            px[2] = (F)((((int)(t.cam * 1000) + i + j + t.n_mat) % 256) / 255.0 * p.samples_per_pixel);
        }
}
} // namespace

struct rrtx_ctx {
    rrtx_params p;
    Tables t;
    bool have = false;
};
struct rrtx_group {
    rrtx_params p;
    int n = 0;
    Tables t;
    bool have = false;
};

extern "C" {
const char *rrtx_last_error(void) { return g_err.c_str(); }
int rrtx_abi_version(void) { return RRTX_ABI_VERSION; }
int rrtx_device_count(void) { return 2; }
int rrtx_query(int device, rrtx_devinfo *out)
{
    if (device < 0 || device >= 2) {
        g_err = "invalid device ordinal";
        return RRTX_E_DEVICE;
    }
    memset(out, 0, sizeof *out);
    strcpy(out->name, "stand-in device (sanitizer build)");
    return RRTX_OK;
}
int rrtx_runtime_version(void) { return 0; }
int rrtx_pin_host(void *, size_t) { return RRTX_OK; }
int rrtx_unpin_host(void *) { return RRTX_OK; }
int rrtx_create(const rrtx_params *params, rrtx_ctx **out)
{
    nap();
    if (params->device < 0 || params->device >= 2) {
        g_err = "invalid device ordinal";
        return RRTX_E_DEVICE;
    }
    *out = new rrtx_ctx{*params, {}, false};
    return RRTX_OK;
}
void rrtx_destroy(rrtx_ctx *c) { delete c; }
static Tables tables_of(const rrtx_scene_desc *s)
{
    Tables t;
    t.n_sph = s->num_spheres, t.n_msph = s->num_moving_spheres, t.n_tri = s->num_triangles, t.n_mat = s->num_materials;
    if (s->camera) {
        if (s->fp64)
            memcpy(&t.cam, s->camera, sizeof(double));
        else {
            float f;
            memcpy(&f, s->camera, sizeof f);
            t.cam = f;
        }
    }
    return t;
}
int rrtx_set_scene(rrtx_ctx *c, const rrtx_scene_desc *s)
{
    nap();
    c->t = tables_of(s), c->have = true;
    return RRTX_OK;
}
int rrtx_render(rrtx_ctx *c, void *fb, rrtx_stats *st)
{
    nap();
    if (!c->have) return RRTX_E_NO_SCENE;
    if (c->p.fp64)
        paint<double>(c->p, c->t, (double *)fb, 0, c->p.image_height);
    else
        paint<float>(c->p, c->t, (float *)fb, 0, c->p.image_height);
    if (st) memset(st, 0, sizeof *st), st->samples = (uint64_t)c->p.image_width * c->p.image_height * c->p.samples_per_pixel;
    return RRTX_OK;
}
int rrtx_group_create(const rrtx_params *params, int n, const int32_t *, int flags, rrtx_group **out)
{
    nap();
    if (n > 2 && !(flags & RRTX_GROUP_REHEARSAL)) {
        g_err = "more members than devices";
        return RRTX_E_DEVICE;
    }
    *out = new rrtx_group{*params, n, {}, false};
    return RRTX_OK;
}
void rrtx_group_destroy(rrtx_group *g) { delete g; }
int rrtx_group_set_scene(rrtx_group *g, const rrtx_scene_desc *s)
{
    nap();
    g->t = tables_of(s), g->have = true;
    return RRTX_OK;
}
int rrtx_group_render(rrtx_group *g, void *fb, rrtx_group_stats *st)
{
    nap();
    if (!g->have) return RRTX_E_NO_SCENE;
    if (g->p.fp64)
        paint<double>(g->p, g->t, (double *)fb, 0, g->p.image_height);
    else
        paint<float>(g->p, g->t, (float *)fb, 0, g->p.image_height);
    if (st) memset(st, 0, sizeof *st), st->n_devices = g->n;
    return RRTX_OK;
}
}
