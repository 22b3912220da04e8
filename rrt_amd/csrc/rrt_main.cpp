// `rrt` / `rrtd`: command-line front end with the flags, stderr chatter, exit codes and output
// formats of the reference's main.cpp, driving the render path through the
// C ABI of include/rrtx.h only.  Built twice: `rrt` (float) and `rrtd` (-DRRTX_DOUBLE).
//
// Flags (recognised by their second character, as main.cpp:69-119 does):
//   -i file.txt  -o file.png  -w W  -h H  -s spp  -d max_depth  -b  -tx N  -ty N  -q  -D device
// Several -i (each with its own -o, paired in order) render a batch of scenes in one process: the device
// context is created once, and while scene k + 1 is parsed, uploaded and rendered, a worker thread
// quantises and encodes scene k (the reference's README lists "input list of
// scenes to render" as an idea, README.md:64).
// Additions of this implementation (no collision with the reference's letters):
//   -C <samples per work item>   (0 = automatic, -1 = one item per pixel: reference sum order)
//   -S <seed>                    (RNG base seed, default 1984)
//   -R <rank> -N <count> -T <tile_rows>   render only one row-tile shard of the frame
#include <unistd.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <future>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rrtx.h"

#ifdef RRTX_DOUBLE
static const int kFp64 = 1;
static const char *kFpName = "double";
typedef double fp_t;
#else
static const int kFp64 = 0;
static const char *kFpName = "float";
typedef float fp_t;
#endif

static void usage(const char *arg)
{
    std::cerr << "Unexpected argument: " << arg << "\n\n";
    std::cerr << "Usage: rrt [options]\n";
    std::cerr << "  -i file.txt         : input scene file (repeat -i / -o pairs to render a batch in one process)\n";
    std::cerr << "  -o file.png         : output raytraced PNG image (default is PPM to stdout)\n";
    std::cerr << "  -w <width>          : output image width. (default = 1200)\n";
    std::cerr << "  -h <height>         : output image height. (800)\n";
    std::cerr << "  -s <samples>        : number of samples per pixel. (10)\n";
    std::cerr << "  -d <max_depth>      : may ray recursion depth. (50)\n";
    std::cerr << "  -b                  : disable acceleration (scan the primitive list for every ray segment).\n";
    std::cerr << "  -tx <num_threads_x> : number of threads per block in x. (8)\n";
    std::cerr << "  -ty <num_threads_y> : number of threads per block in y. (8)\n";
    std::cerr << "  -q                  : query devices & HIP info\n";
    std::cerr << "  -D <device number>  : use this HIP device (0)\n";
    std::cerr << "  -C <chunk>          : samples per work item (0 = auto, -1 = whole pixel)\n";
    std::cerr << "  -S <seed>           : random seed (1984)\n";
    std::cerr << "  -R <rank> -N <count> -T <tile_rows> : render one row-tile shard only\n";
    std::exit(1);
}

static void die_device(int rc)
{
    // check_cuda, rrt.cu:31-40
    std::cerr << "HIP error = " << rc << " : " << rrtx_last_error() << "\n";
    std::exit(99);
}

static void query_devices()
{
    // main.cpp:14-30 (stdout, like the reference)
    int count = rrtx_device_count();
    for (int i = 0; i < count; i++) {
        rrtx_devinfo p;
        int rc = rrtx_query(i, &p);
        if (rc) die_device(rc);
        std::cout << "hipGetDeviceProperties #" << i << "\n";
        std::cout << "  name                        " << p.name << "\n";
        std::cout << "  major.minor                 " << p.major << "." << p.minor << "\n";
        std::cout << "  multiProcessorCount         " << p.multi_processor_count << "\n";
        std::cout << "  sharedMemPerBlock           " << p.shared_mem_per_block << "\n";
        std::cout << "  maxThreadsPerBlock          " << p.max_threads_per_block << "\n";
        std::cout << "  maxThreadsPerMultiProcessor " << p.max_threads_per_multiprocessor << "\n";
        std::cout << "  unifiedAddressing           " << p.unified_addressing << "\n";
        std::cout << "  l2CacheSize                 " << p.l2_cache_size << "\n";
    }
}

int main(int argc, char *argv[])
{
    rrtx_params prm;
    std::memset(&prm, 0, sizeof prm);
    prm.image_width = 1200;
    prm.image_height = 800;
    prm.samples_per_pixel = 10;
    prm.max_depth = 50;
    prm.use_bvh = 1;
    prm.threads_x = 8;
    prm.threads_y = 8;
    prm.fp64 = kFp64;
    prm.shard_count = 1;
    prm.tile_rows = 4;
    prm.collect_stats = 1;
    std::vector<std::string> scene_files;
    std::vector<const char *> png_files;

    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] != '-') usage(argv[i]);
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) usage(argv[i]);
            return argv[++i];
        };
        switch (argv[i][1]) {
        case 'i': scene_files.push_back(next()); break;
        case 'o': png_files.push_back(next()); break;
        case 'w': prm.image_width = atoi(next()); break;
        case 'h': prm.image_height = atoi(next()); break;
        case 's': prm.samples_per_pixel = atoi(next()); break;
        case 'd': prm.max_depth = atoi(next()); break;
        case 'b': prm.use_bvh = 0; break;
        case 't':
            if (argv[i][2] == 'x')
                prm.threads_x = atoi(next());
            else if (argv[i][2] == 'y')
                prm.threads_y = atoi(next());
            else
                usage(argv[i]);
            break;
        case 'q': query_devices(); break;
        case 'D': {
            // main.cpp:107-110 selects the device while parsing (cudaSetDevice, fatal on a bad ordinal): the same
            // here - a bad ordinal ends the run with code 99 before any scene is read
            prm.device = atoi(next());
            rrtx_devinfo probe;
            int rc = rrtx_query(prm.device, &probe);
            if (rc) die_device(rc);
            break;
        }
        case 'C': prm.sample_chunk = atoi(next()); break;
        case 'S': prm.seed = (uint32_t)strtoul(next(), nullptr, 10); break;
        case 'R': prm.shard_rank = atoi(next()); break;
        case 'N': prm.shard_count = atoi(next()); break;
        case 'T': prm.tile_rows = atoi(next()); break;
        default: usage(argv[i]);
        }
    }

    if (scene_files.empty()) {
        std::cerr << "ERROR: no scene loaded." << std::endl;
        return 1;
    }

    rrtx_ctx *ctx = nullptr;
    std::shared_ptr<std::vector<fp_t>> frames[2]; // page-locked once, used in turn (see below)
    std::future<int> writer; // the previous scene's quantise + encode, running beside this scene's render
    int exit_code = 0;
    for (size_t job = 0; job < scene_files.size(); ++job) {
        const std::string &scene_file = scene_files[job];
        const char *png_file = job < png_files.size() ? png_files[job] : nullptr;
        rrtx_scene *scene = nullptr;
        int rc = rrtx_scene_load(scene_file.c_str(), prm.image_width, prm.image_height, kFp64, &scene);
        if (rc) {
            if (writer.valid()) writer.get();
            for (auto &f : frames)
                if (f) (void)rrtx_unpin_host(f->data());
            int code = rrtx_scene_exit_code();
            return code ? code : 1;
        }
        int32_t counts[6];
        rrtx_scene_counts(scene, counts);
        rrtx_scene_desc desc;
        rrtx_scene_describe(scene, &desc);
        // scene.h:443-451
        std::cerr << "read scene file: " << scene_file << "\n";
        std::cerr << "material count:  " << counts[0] << "\n";
        std::cerr << "sphere count:    " << counts[1] << std::endl;
        std::cerr << "msphere count:   " << counts[2] << std::endl;
        std::cerr << "obj count:       " << counts[4] << std::endl;
        std::cerr << "obj_inst count:  " << counts[5] << std::endl;
        {
            const fp_t *cam = (const fp_t *)desc.camera;
            if (cam[22] != cam[23]) std::cerr << "camera time:     " << cam[22] << " - " << cam[23] << std::endl;
        }

        std::time_t render_time = std::time(nullptr);
        std::tm render_tm = *std::localtime(&render_time);

        if (!ctx) {
            rc = rrtx_create(&prm, &ctx);
            if (rc) die_device(rc);
        }
        rc = rrtx_set_scene(ctx, &desc);
        if (rc) die_device(rc);

        // two frame buffers, used in turn: one is being quantised and encoded by the writer task while the next
        // scene renders into the other (allocating and zero-filling 11 MB per frame cost more than a short render)
        std::shared_ptr<std::vector<fp_t>> &slot = frames[job & 1];
        if (!slot) {
            slot = std::make_shared<std::vector<fp_t>>((size_t)prm.image_width * prm.image_height * 3, (fp_t)0);
            (void)rrtx_pin_host(slot->data(), slot->size() * sizeof(fp_t)); // (best effort: unpinned it is only slower)
        }
        auto fb = slot;
        // rrt.cu:195-202,261
        std::cerr << "HIP Runtime Version " << rrtx_runtime_version() << "\n";
        std::cerr << "Rendering a " << prm.image_width << "x" << prm.image_height << " image with " << prm.samples_per_pixel << " samples per pixel ";
        rrtx_stats st;
        std::memset(&st, 0, sizeof st);
        std::cerr << "(" << kFpName << (prm.use_bvh ? ", acceleration grid where the scene allows" : ", list scan") << ").\n";
        std::cerr << "num_hittables = " << (counts[1] + counts[2] + counts[3]) << "\n";
        std::cerr << "HIP Device: " << prm.device << std::endl;

        rc = rrtx_render(ctx, fb->data(), &st);
        if (rc) die_device(rc);

        const double seconds = st.kernel_ms / 1000.0;
        std::cerr << "took " << seconds << " seconds.\n";
        char hostname[HOST_NAME_MAX + 1];
        hostname[0] = 0;
        gethostname(hostname, sizeof hostname);
        char when[128];
        std::strftime(when, sizeof when, "%c %Z,", &render_tm);
        // rrt.cu:312-315: stats,<date>,<host>,<runtime>,<fp>,w,h,spp,blocks,tx,ty,seconds
        std::cerr << "stats," << when << hostname << ",HIP" << rrtx_runtime_version() << "," << kFpName << "," << prm.image_width << "," << prm.image_height << ","
                  << prm.samples_per_pixel << "," << st.grid_blocks << "," << prm.threads_x << "," << prm.threads_y << "," << seconds << "\n";
        if (seconds > 0)
            std::cerr << "rate," << (double)st.samples / seconds / 1e6 << " Msamples/s," << st.segments << " segments," << st.prim_tests << " primitive tests,"
                      << (double)st.bytes_algorithmic / seconds / 1e9 << " GB/s algorithmic," << (st.accel_cells ? "grid of " + std::to_string(st.accel_cells) + " cells" : std::string("list scan")) << "\n";
        rrtx_scene_free(scene);

        // main.cpp:140-162: quantise, flip, write — off the render path: it overlaps the next scene
        if (writer.valid() && writer.get()) exit_code = 1; // (keeps the outputs in order, PPMs on stdout included)
        const int w = prm.image_width, h = prm.image_height, spp = prm.samples_per_pixel;
        writer = std::async(std::launch::async, [fb, png_file, w, h, spp]() -> int {
            std::vector<uint8_t> rgb((size_t)w * h * 3);
            rrtx_quantise(fb->data(), kFp64, w, h, spp, rgb.data());
            const int wrc = png_file == nullptr ? rrtx_write_ppm(nullptr, rgb.data(), w, h) : rrtx_write_png(png_file, rgb.data(), w, h);
            if (wrc) std::cerr << "ERROR: could not write image\n";
            return wrc;
        });
    }
    if (writer.valid() && writer.get()) exit_code = 1;
    for (auto &f : frames)
        if (f) (void)rrtx_unpin_host(f->data());
    if (ctx) rrtx_destroy(ctx);
    return exit_code;
}
