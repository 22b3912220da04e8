/*
 * rrtx.h — C ABI of the MI355X-native render path for rogerallen/rrt.
 *
 * This is the drop-in boundary for the reference's ONE hot path: everything `Rrt::render`
 * does on the device (rrt.cu:186-334) and, on the host side of it, the three things its only
 * caller needs (main.cpp:123-167): a scene parsed into POD tables, the framebuffer, and the
 * 8-bit quantiser / image writers.  Plain pointers and sizes only — no C++ types, no torch
 * types, no exceptions cross this boundary; every entry point returns 0 or a negative
 * RRTX_E_* code (host callers map non-zero to "print + exit(99)", the behaviour of
 * check_cuda, rrt.cu:31-40).  All file:line citations are into the reference repository.
 *
 * The table structs below are LAYOUT-IDENTICAL to what the reference marshals to its
 * create_world kernel (rrt.cu:124-128, 217-247): scene.h:27-54 (triangle / sphere / moving
 * sphere), scene.h:183-208 (material) and camera.h:43-48 (camera), in both FP_T = float
 * (`rrt`) and FP_T = double (`rrtd`) builds.  A reference maintainer can therefore hand the
 * arrays rrt.cu already builds straight to rrtx_set_scene().
 */
#ifndef RRTX_H
#define RRTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRTX_VERSION_STRING "rrtx 0.4 (gfx950)"
/* Layout version of the structs below.  rrtx_stats and rrtx_group_stats are filled with sizeof(the library's struct) bytes:
 * a host compiled against a header of another RRTX_ABI_VERSION must not pass its own structs.  Check once at start-up:
 *     if (rrtx_abi_version() != RRTX_ABI_VERSION) refuse;
 * History: 3 = round 3 (rrtx_stats grew walk_cells / walk_pairs, rrtx_group_stats the RCCL fields, without a bump: that
 * was a silent break); 4 = rrtx_stats.convergence_faults, sky_pixels, first_bounce. */
#define RRTX_ABI_VERSION 4

/* ---- error codes ------------------------------------------------------------------- */
#define RRTX_OK 0
#define RRTX_E_INVALID (-1)   /* bad argument / inconsistent parameters                    */
#define RRTX_E_DEVICE (-2)    /* HIP runtime error (message in rrtx_last_error())          */
#define RRTX_E_NO_SCENE (-3)  /* render called before rrtx_set_scene                       */
#define RRTX_E_IO (-4)        /* file could not be opened / written                        */
#define RRTX_E_PARSE (-5)     /* scene file rejected; rrtx_scene_exit_code() has the
                                 reference's exit code (1..4)                              */
#define RRTX_E_UNSUPPORTED (-6)

/* ---- POD tables (reference layouts) ------------------------------------------------ */
enum { RRTX_LAMBERTIAN = 0, RRTX_METAL = 1, RRTX_DIELECTRIC = 2 }; /* scene.h:181 */

/* camera.h:43-48: origin, lower_left_corner, horizontal, vertical, u, v, w, lens_radius,
 * time0, time1 — 24 FP_T, 96 B (float) / 192 B (double). */
typedef struct rrtx_camera_f32 {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius, time0, time1;
} rrtx_camera_f32;
typedef struct rrtx_camera_f64 {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius, time0, time1;
} rrtx_camera_f64;

/* scene.h:183-208: enum type @0, union @8; 32 B (float) / 40 B (double). */
typedef struct rrtx_material_f32 {
    int32_t type;
    union {
        struct { float albedo[3]; } lambertian;
        struct { float albedo[3]; double fuzz; } metal;
        struct { double ref_idx; } dielectric;
    } mat;
} rrtx_material_f32;
typedef struct rrtx_material_f64 {
    int32_t type;
    union {
        struct { double albedo[3]; } lambertian;
        struct { double albedo[3]; double fuzz; } metal;
        struct { double ref_idx; } dielectric;
    } mat;
} rrtx_material_f64;

/* scene.h:43-47: 32 B (float; radius @16, material_idx @24) / 40 B (double). */
typedef struct rrtx_sphere_f32 { float center[3]; double radius; int32_t material_idx; } rrtx_sphere_f32;
typedef struct rrtx_sphere_f64 { double center[3]; double radius; int32_t material_idx; } rrtx_sphere_f64;

/* scene.h:49-54: 56 B (float) / 80 B (double). */
typedef struct rrtx_moving_sphere_f32 { float center0[3], center1[3]; double time0, time1, radius; int32_t material_idx; } rrtx_moving_sphere_f32;
typedef struct rrtx_moving_sphere_f64 { double center0[3], center1[3]; double time0, time1, radius; int32_t material_idx; } rrtx_moving_sphere_f64;

/* scene.h:27-41 scene_instance_triangle (post-transform, flattened): 40 B / 80 B. */
typedef struct rrtx_triangle_f32 { float vertices[3][3]; int32_t material_idx; } rrtx_triangle_f32;
typedef struct rrtx_triangle_f64 { double vertices[3][3]; int32_t material_idx; } rrtx_triangle_f64;

/* What crosses to the device — the argument list of create_world (rrt.cu:124-128).  All
 * pointers are caller-owned host memory, read during rrtx_set_scene() only. */
typedef struct rrtx_scene_desc {
    int32_t fp64;               /* 0: the *_f32 structs above, 1: the *_f64 ones            */
    const void *camera;
    int32_t num_materials;
    const void *materials;
    int32_t num_spheres;
    const void *spheres;
    int32_t num_moving_spheres;
    const void *moving_spheres;
    int32_t num_triangles;
    const void *triangles;
} rrtx_scene_desc;

/* ---- render context ------------------------------------------------------------------ */
/* Constructor arguments of class Rrt (rrt.h:16-31) plus what the MI355X path adds. */
typedef struct rrtx_params {
    int32_t image_width, image_height; /* -w / -h                                          */
    int32_t samples_per_pixel;         /* -s                                               */
    int32_t max_depth;                 /* -d                                               */
    int32_t use_bvh;                   /* -b clears it (main.cpp:67,90).  Non-zero: segments are resolved
                                          through an acceleration grid when the scene allows one
                                          (rrtx_stats.accel_cells); zero: the hittable_list scan.
                                          Same image either way, bit for bit - except fp32 triangle
                                          meshes, see RRTX_FLAG_EXACT_ACCEL / rrtx_stats.accel_exact.  */
    int32_t threads_x, threads_y;      /* -tx / -ty: accepted; reported in the stats line    */
    int32_t fp64;                      /* 0 = `rrt` (float), 1 = `rrtd` (double)            */
    int32_t device;                    /* HIP device ordinal (-D)                           */
    uint32_t seed;                     /* RNG base seed; 0 selects 1984 (cf. rrt.cu:88)     */
    int32_t sample_chunk;              /* samples per work item; 0 = automatic, >= spp (or
                                          -1) = one item per pixel = the reference's own
                                          summation order                                   */
    /* Row-tile sharding of the frame over `shard_count` devices/processes: rows are cut into
     * tiles of `tile_rows`, tile t belongs to shard (t mod shard_count).  count 0/1 = whole
     * frame.  The image is bit-identical for every (shard_count, tile_rows).               */
    int32_t shard_rank, shard_count, tile_rows;
    int32_t collect_stats;             /* 1: count segments / primitive tests on the device  */
    int32_t flags;                     /* RRTX_FLAG_* bits                                   */
    int32_t handoff_lanes;             /* tuning: a wave parks its unfinished items for the tail
                                          kernel once the queue is dry and at most this many of
                                          its 64 lanes are alive; 0 = default (7)              */
    int32_t handoff_iters;             /* tuning: ... or this many iterations after the queue ran
                                          dry, whichever comes first; 0 = default (20; 4 for scenes
                                          with triangles / moving spheres under use_bvh)        */
    int32_t list_passes;               /* tuning: camera-ray LIST passes allowed between two SCAN
                                          passes; 0 = default (3), -1 = none                    */
    int32_t taper_samples;             /* tuning: this many samples at the end of the work queue are
                                          handed out one by one instead of in chunks (rounded up to
                                          whole pixels); 0 = automatic (currently none), -1 = none.
                                          Scheduling only: the image does not depend on it       */
    int32_t reserved[1];
} rrtx_params;

/* Scan every sphere with the reference's own discriminant (18 VALU ops per test) instead of the
 * conservative 8-op FMA filter + exact refinement.  Both produce bit-identical images; the flag
 * exists for A/B measurements and is forced internally for scenes whose magnitudes fall outside
 * the filter's proven range. */
#define RRTX_FLAG_EXACT_SCAN 1
/* Where the scan reads sphere records from (A/B switches; default = alternate between scalar
 * loads and an LDS copy whenever the table fits in LDS). */
#define RRTX_FLAG_SCAN_SCALAR_ONLY 2
#define RRTX_FLAG_SCAN_LDS_ONLY 4
/* Let the render kernel finish every path itself instead of handing the last few to the tail
 * kernel (A/B switch; the images are identical). */
#define RRTX_FLAG_NO_TAIL_KERNEL 8
/* Send camera rays through the full scan too instead of their pixel's candidate list (A/B switch). */
#define RRTX_FLAG_NO_PRIMARY_LISTS 16
/* Test mode: every camera ray intersected through its list is also scanned sequentially on its
 * lane; rrtx_stats.list_mismatches counts the rays for which the two disagree (must stay 0).
 * With use_bvh the same is done for every segment resolved through the acceleration grid. */
#define RRTX_FLAG_VERIFY_LISTS 32
/* use_bvh in fp32 with triangle meshes: keep the list scan's bits at any price.  Without this flag a mesh is entered
 * into the acceleration grid under an empirical inflation (rrtx_grid.h, kApproxTriInflation): the reference's
 * triangle test (triangle.h:38-75), evaluated in fp32, reports hits on rays that graze a triangle's plane from far
 * outside it, no finite inflation is PROVEN to catch them all, and a ray segment in ~10^-5 (measured, bounded by
 * tests/test_gpu_mesh.py) then resolves differently from the sequential scan - exactly where the reference's own BVH
 * (its default; bvh.h:167-175) differs from its own `-b` list scan.  rrtx_stats.accel_exact says which rule is in use;
 * spheres, moving spheres and fp64 meshes are always under the proven rule.  With this flag fp32 triangles stay
 * outside the grid (always-list of <= 48, else the whole scene is scanned: O(n) per segment). */
#define RRTX_FLAG_EXACT_ACCEL 64
/* List scan (use_bvh = 0): finish the paths a launch parks at its end with the tail kernel (8 lanes per ray on the
 * list) instead of a resume pass through the acceleration grid (A/B switch; the images are identical - the grid is
 * only used for this where it is proven to reproduce the scan bit for bit). */
#define RRTX_FLAG_NO_TAIL_GRID 128
/* List scan of a scene of spheres alone: keep the conservative filter on the vector unit (7 FMAs per ray and sphere)
 * instead of the matrix cores (two chained v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays; A/B switch, identical images). */
#define RRTX_FLAG_SCAN_NO_MFMA 256

/* sample_chunk = -1 (the reference's own order of summation): schedule it as ONE work item per pixel, as up to ABI 3, instead of small work items
 * that store every sample for a final running sum (A/B switch; the images are identical, the one-item schedule takes about twice the time). */
#define RRTX_FLAG_ONE_ITEM_PER_PIXEL 512

/* Keep the pixels whose camera-ray candidate list is empty - sky in every sample - in the render kernel's queue instead of finishing their work items in a dense
 * kernel of their own (A/B switch; the images are identical). */
#define RRTX_FLAG_NO_SKY_SPLIT 1024
/* Launch that kernel on the render's own stream, ahead of the render kernel, as round 4 first had it, instead of beside it on a low-priority stream of the
 * context's own, where it fills the end of the launch (A/B switch; the images are identical). */
#define RRTX_FLAG_SKY_SAME_STREAM 8192

/* The first bounce of every sample - camera ray, closest hit among the pixel's candidates, scatter - as a dense pre-pass of each launch that leaves one record per
 * sample, instead of inside the render loop (LIST passes).  Used by itself where it was measured to pay: use_bvh, scenes of spheres alone, launches of 16 M samples
 * and more, and 8 values of the frame's precision per sample of device memory to spare (15 GB for 1200x800 spp 500 in fp32); skipped silently otherwise.
 * NO_FIRST_BOUNCE never uses it, FIRST_BOUNCE_ALWAYS uses it for every use_bvh launch of a scene of spheres alone that has the memory, whatever its size (the
 * list scan never does: measured to gain nothing there, its kernels are compiled without the record path): A/B switches, the images are identical. */
#define RRTX_FLAG_NO_FIRST_BOUNCE 2048
#define RRTX_FLAG_FIRST_BOUNCE_ALWAYS 4096

typedef struct rrtx_stats {
    double kernel_ms;        /* HIP-event time of the render (+finalise) kernels of the LAST
                                render, measured on the stream they were launched on         */
    double kernel_ms_sum;    /* the same, summed over the `renders` launches since the
                                previous rrtx_collect / rrtx_render                          */
    int32_t renders;
    int32_t accel_cells;     /* use_bvh: cells of the acceleration grid in use, 0 = the list is scanned   */
    double wall_ms;          /* host wall time of the last blocking render call              */
    uint64_t samples;        /* pixels_rendered * spp                                        */
    uint64_t segments;       /* path segments traced (0 unless collect_stats)                */
    uint64_t prim_tests;     /* primitive intersection tests = segments * num_primitives     */
    uint64_t bytes_algorithmic; /* prim_tests * B_prim + framebuffer bytes     */
    int32_t grid_blocks, block_threads;
    int32_t sample_chunk;    /* the value actually used                                      */
    int32_t local_rows;      /* rows rendered by this shard                                  */
    uint64_t candidates;     /* (ray, primitive) pairs that reached the exact refinement      */
    int32_t scan_filter;     /* 1 if the conservative scan filter was used                    */
    int32_t list_mismatches; /* RRTX_FLAG_VERIFY_LISTS: camera rays whose list hit != scan hit */
    uint64_t scanned_segments; /* segments that went through the full primitive scan (the others
                                  are camera rays resolved from their pixel's candidate list)  */
    int32_t accel_exact;     /* 1: the closest hit in use is proven to equal the list scan's bit for bit (always, unless
                                fp32 triangles were entered into the grid under the approximate rule: 0)              */
    int32_t scan_mfma;       /* 1: the list scan's filter ran on the matrix cores (scenes of spheres alone whose f16 operands fit LDS;
                                RRTX_FLAG_SCAN_NO_MFMA keeps it on the vector unit - the images are identical)                */
    uint64_t walk_cells;     /* grid cells the walks stepped through, and ...                                              */
    uint64_t walk_pairs;     /* ... (ray, entry) pairs they tested - counted by the densely pairing variants (scenes with
                                triangles / moving spheres under use_bvh) only, 0 elsewhere                                */
    uint64_t convergence_faults; /* waves that reached a wave-wide step (matrix-core scan, dense pairing) with lanes masked off: must
                                be 0 - the kernels count it instead of assuming it (always collected)                      */
    int32_t sky_pixels;      /* pixels of this shard whose work items a dense kernel of camera rays finished (their candidate list is empty: sky in
                                every sample); 0 = no split (RRTX_FLAG_NO_SKY_SPLIT, scenes with anything but spheres, no such pixel)     */
    int32_t first_bounce;    /* 1: the first bounce of every queued sample was a dense pre-pass of the launch (RRTX_FLAG_NO_FIRST_BOUNCE /
                                RRTX_FLAG_FIRST_BOUNCE_ALWAYS; by itself: use_bvh, spheres alone, 16 M samples and more, memory to spare)  */
} rrtx_stats;

typedef struct rrtx_devinfo { /* the fields main.cpp:14-30 prints for -q */
    char name[256];
    int32_t major, minor;
    int32_t multi_processor_count;
    int64_t shared_mem_per_block;
    int32_t max_threads_per_block;
    int32_t max_threads_per_multiprocessor;
    int32_t unified_addressing;
    int32_t l2_cache_size;
    int64_t total_global_mem;
    int32_t clock_khz;
} rrtx_devinfo;

typedef struct rrtx_ctx rrtx_ctx;

const char *rrtx_version(void);
int rrtx_abi_version(void); /* the RRTX_ABI_VERSION the library was built with */
/* Thread-local description of the last failure on the calling thread ("" if none). */
const char *rrtx_last_error(void);

int rrtx_device_count(void);                       /* main.cpp:17 cudaGetDeviceCount        */
int rrtx_query(int device, rrtx_devinfo *out);     /* main.cpp:19 cudaGetDeviceProperties   */
int rrtx_runtime_version(void);                    /* rrt.cu:195 cudaRuntimeGetVersion      */
/* Optional: page-lock a caller-owned frame buffer so that rrtx_render's copy-back runs at link speed (the
 * reference returns managed memory, rrt.cu:204; a pageable 11 MB frame costs ~1.5 ms more per render).
 * rrtx_unpin_host before the memory is freed.  Both are no-ops returning RRTX_OK on NULL / 0 bytes. */
int rrtx_pin_host(void *ptr, size_t bytes);
int rrtx_unpin_host(void *ptr);

/* Rrt::Rrt (rrt.h:16-31). */
int rrtx_create(const rrtx_params *params, rrtx_ctx **out);
/* Rrt::~Rrt (rrt.cu:336-342). Safe on NULL. */
void rrtx_destroy(rrtx_ctx *ctx);

/* The marshalling half of Rrt::render (rrt.cu:217-270): packs the tables into the device
 * layout and uploads them.  Replaces create_world<<<1,1>>>.  May be called again to swap
 * scenes on a live context (it waits for THIS context's renders to finish first - the stop event of
 * every render enqueued through it, on whatever stream, and its own stream - not for the device:
 * other contexts and a host application's streams keep running; the device buffers of the previous
 * scene are reused and only grow: 0.3 ms per call for final.txt at 1280x720). */
int rrtx_set_scene(rrtx_ctx *ctx, const rrtx_scene_desc *scene);

/* Rows of the frame this context renders (global row numbers, ascending; row 0 = bottom of
 * the image, rrt.cu:117).  Returns the count; fills at most `cap` entries when rows != NULL. */
int rrtx_shard_rows(const rrtx_ctx *ctx, int32_t *rows, int cap);

/* The launch half of Rrt::render (rrt.cu:286-298) + copy-back.  `fb` is caller-owned host
 * memory of image_width*image_height*3 FP_T (float or double per params.fp64), pixel (i,j) at
 * (j*image_width+i)*3, row 0 = bottom, un-normalised sum over samples — exactly the buffer
 * Rrt::render returns (rrt.cu:118).  Only this shard's rows are written.  Blocking. */
int rrtx_render(rrtx_ctx *ctx, void *fb, rrtx_stats *stats);

/* Same render, output left in device memory: `d_rows` is a device pointer to
 * rrtx_shard_rows()*image_width*3 FP_T, local row k = k-th row of rrtx_shard_rows().
 * `hip_stream` is a hipStream_t and is used as given (NULL = HIP's null stream, as in every HIP
 * call; rrtx_stream() returns the context's own non-blocking stream).  Asynchronous: returns
 * after enqueueing; the framebuffer is complete when the stream reaches this point.  This is
 * the entry point the multi-GPU gather and bench.py use (inputs resident in HBM). */
int rrtx_render_device(rrtx_ctx *ctx, void *d_rows, void *hip_stream);
/* The context's own stream (hipStream_t), the one rrtx_render() launches on. */
void *rrtx_stream(rrtx_ctx *ctx);

/* Waits for the renders enqueued so far and fills `stats` for the last one. */
int rrtx_collect(rrtx_ctx *ctx, rrtx_stats *stats);

/* ---- one frame over several devices of one node ------------------------------------------
 * BASELINE.json north_star: "the framebuffer is row-tile-partitioned across the 8 GPUs of one node
 * with a final RCCL gather over xGMI".  The reference has nothing to cite here: it renders on one
 * device (main.cpp:107-110 merely calls cudaSetDevice).  A group is ONE process driving N devices
 * from one host thread: N contexts as above (shard_rank = position in the device list,
 * shard_count = N, params->tile_rows), one stream per device, one RCCL communicator per device
 * (ncclCommInitAll, rccl.h:236).  rrtx_group_render() enqueues the N shard renders, then ONE grouped
 * exchange - every rank ncclSend()s its compact row block to rank 0, which ncclRecv()s them side by
 * side (rccl.h:700; all seven peers of an MI355X have their own xGMI link to the root, so they
 * transmit concurrently) - a de-interleave pass on rank 0 puts the rows where they belong, and one
 * copy brings the frame to the host.  The image is the single-device image, bit for bit, for every
 * N and tile height (the RNG is keyed by the global pixel index).  librccl.so is loaded on first use
 * (dlopen): single-device runs never touch it. */
typedef struct rrtx_group rrtx_group;

/* Several contexts may share a device (the list repeats an ordinal): a REHEARSAL of the N-way
 * decomposition on fewer GPUs, for tests - RCCL cannot build a communicator over duplicates, the row
 * blocks then move with device-to-device copies instead.  Rejected without this flag. */
#define RRTX_GROUP_REHEARSAL 1
/* RCCL greets on STDOUT when a communicator is built ("RCCL version : ..."), and stdout is where `rrt` prints its PPM
 * (main.cpp:142).  Up to ABI 3 rrtx_group_create pointed file descriptor 1 at stderr while ncclCommInitAll ran unless this
 * flag was passed; a library should not touch a process-wide descriptor, so since ABI 4 it never does: a host that owns
 * stdout diverts it itself around rrtx_group_create (rrt_main.cpp does, before any writer task exists; so does
 * rrt_amd.RrtGroup).  The flag is still accepted, and ignored. */
#define RRTX_GROUP_KEEP_STDOUT 2

typedef struct rrtx_group_stats {
    int32_t n_devices;
    int32_t rccl;            /* 1: the row blocks moved through ncclSend / ncclRecv             */
    double render_ms;        /* slowest device's kernel time (HIP events, as rrtx_stats.kernel_ms) */
    double device_ms;        /* first launch -> assembled frame on device 0 (events on its stream): render + gather + de-interleave */
    double gather_ms;        /* device_ms - render_ms: what the exchange and the de-interleave add to the slowest render */
    double wall_ms;          /* host wall time of the call, copy-back included                  */
    double kernel_ms[16];    /* per device (first 16)                                           */
    uint64_t samples, segments, prim_tests, bytes_algorithmic; /* summed over the devices      */
    uint64_t gathered_bytes; /* bytes that crossed to rank 0 (its own block included)           */
    int32_t sample_chunk, accel_cells;
    int32_t accel_exact;     /* as rrtx_stats.accel_exact, 0 if any member renders under the approximate rule */
    int32_t rccl_version;    /* ncclGetVersion() of the library the communicators were built with (0: none) */
    int32_t rccl_comms;      /* communicators alive in this group (= n_devices, 0 in a rehearsal)         */
    int32_t devices[16];     /* HIP ordinal of each member (first 16)                                    */
    int32_t reserved;
} rrtx_group_stats;

/* params: as for rrtx_create; device / shard_rank / shard_count are ignored (set per member).
 * devices == NULL: ordinals 0 .. n_devices-1. */
int rrtx_group_create(const rrtx_params *params, int n_devices, const int32_t *devices, int flags, rrtx_group **out);
void rrtx_group_destroy(rrtx_group *g);
int rrtx_group_size(const rrtx_group *g);
/* The i-th member (borrowed; e.g. for rrtx_shard_rows). */
rrtx_ctx *rrtx_group_member(rrtx_group *g, int i);
int rrtx_group_set_scene(rrtx_group *g, const rrtx_scene_desc *scene);
/* `fb`: the whole frame, as rrtx_render's.  Blocking. */
int rrtx_group_render(rrtx_group *g, void *fb, rrtx_group_stats *stats);
/* Same, the frame left on the group's first device (pointer valid until the next render / destroy). */
int rrtx_group_render_device(rrtx_group *g, void **d_frame, rrtx_group_stats *stats);

/* ---- host side of the seam: scene parser (scene.h:212-452) ----------------------------- */
typedef struct rrtx_scene rrtx_scene;
/* Parses `path` for a w x h frame (the camera's aspect ratio comes from them, scene.h:254).
 * On a rejected file returns RRTX_E_PARSE / RRTX_E_IO and *out = NULL. */
int rrtx_scene_load(const char *path, int image_width, int image_height, int fp64, rrtx_scene **out);
/* The reference's process exit code for the last rrtx_scene_load failure on this thread
 * (scene.h:222,289,433-441: 1 = obj errors, 2 = cannot open, 3 = unknown material, 4 = scene
 * sanity). */
int rrtx_scene_exit_code(void);
/* The message rrtx_scene_load printed on stderr for that failure (the reference's own text, scene.h:222,289,433-441;
 * "" if the last load on this thread succeeded).  rrtx_scene_load_quiet() is rrtx_scene_load() without the print:
 * for callers that parse on a helper thread and want the message in their own log, in order (`rrt` batches). */
const char *rrtx_scene_error(void);
int rrtx_scene_load_quiet(const char *path, int image_width, int image_height, int fp64, rrtx_scene **out);
void rrtx_scene_free(rrtx_scene *s);
/* Borrowed view of the parsed tables, valid until rrtx_scene_free. */
int rrtx_scene_describe(const rrtx_scene *s, rrtx_scene_desc *out);
/* counts[6]: materials, spheres, moving spheres, instance triangles, objs, obj instances
 * (the numbers scene.h:443-448 prints). */
int rrtx_scene_counts(const rrtx_scene *s, int32_t counts[6]);

/* ---- host side of the seam: output (color.h:8-32, main.cpp:140-167) -------------------- */
/* 8-bit quantiser: fb (row 0 = bottom) -> rgb (top row first), w*h*3 bytes. */
int rrtx_quantise(const void *fb, int fp64, int image_width, int image_height, int samples_per_pixel, uint8_t *rgb);
/* ASCII PPM "P3" exactly as main.cpp:142-148 prints it; path NULL or "-" = stdout. */
int rrtx_write_ppm(const char *path, const uint8_t *rgb, int image_width, int image_height);
/* 8-bit RGB PNG (decoded pixels identical to the reference's stbi_write_png output). */
int rrtx_write_png(const char *path, const uint8_t *rgb, int image_width, int image_height);

#ifdef __cplusplus
}
#endif
#endif /* RRTX_H */
