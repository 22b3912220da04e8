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
//   -G <gpus>                    several devices of the node, one process: a frame is cut into row tiles over them and
//                                gathered with RCCL (rrtx_group); a batch of >= gpus scenes is dealt out scene by scene
//   -E                           rehearsal of -G on fewer devices than members (they share what is there)
//   -X                           exact acceleration: fp32 triangle meshes stay out of the grid (RRTX_FLAG_EXACT_ACCEL), so that
//                                the image equals `-b`'s bit for bit also for them (spheres and fp64 meshes always do)
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <sstream>
#include <thread>

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
    std::cerr << "  -G <gpus>           : use this many devices, starting at -D: a frame is cut into row tiles (-T) and gathered\n";
    std::cerr << "                        over RCCL; a batch of at least as many scenes is dealt out scene by scene instead\n";
    std::cerr << "  -E                  : rehearsal: let the -G members share the devices present (tests)\n";
    std::cerr << "  -X                  : exact acceleration: fp32 triangle meshes are not gridded (same bits as -b; slower)\n";
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


// ---------------------------------------------------------------------------------------------
// jobs and workers
// ---------------------------------------------------------------------------------------------
struct Job {
    std::string scene_file;
    const char *png_file = nullptr;
    size_t ppm_index = (size_t)-1; // position among the jobs that print a PPM to stdout (they must come out in order)
};

struct Shared {
    const std::vector<Job> *jobs = nullptr;
    std::atomic<size_t> next_job{0};
    std::atomic<int> exit_code{0};
    std::atomic<bool> stop{false};
    std::mutex log_mu; // a job's stderr lines go out in one piece
    std::mutex out_mu; // stdout: PPMs in job order
    std::condition_variable out_cv;
    size_t next_ppm = 0;
};

struct Worker {
    rrtx_params prm;
    Shared *shared = nullptr;
    int group_size = 0; // > 1: this worker drives a group of devices (row-tile shards of every frame)
    bool group_rehearsal = false;
    std::vector<int32_t> group_devices;
    rrtx_ctx *ctx = nullptr;
    rrtx_group *group = nullptr;
    // kWriters + 1 frame buffers, page-locked once and used in turn, and up to kWriters writer tasks (quantise + encode +
    // write) in flight beside this scene's render: a noisy 1280x720 frame takes a writer 4.6 - 5.2 ms on 8 threads (spp 1:
    // 11 ms), a render at the reference's animation settings 5.2 ms - with ONE writer a batch can run at the writer's pace
    static constexpr int kWriters = 2;
    std::shared_ptr<std::vector<fp_t>> frames[kWriters + 1];
    std::future<int> writers[kWriters]; // writers[k % kWriters] works on frames[k % (kWriters + 1)]
    size_t rendered = 0;

    void log(const std::string &text)
    {
        std::lock_guard<std::mutex> lock(shared->log_mu);
        std::cerr << text << std::flush;
    }
    [[noreturn]] void die(int rc, const std::string &so_far)
    {
        // check_cuda, rrt.cu:31-40: fatal.  (_Exit: other workers may be in the middle of a render)
        {
            std::lock_guard<std::mutex> lock(shared->log_mu);
            std::cerr << so_far << "HIP error = " << rc << " : " << rrtx_last_error() << "\n" << std::flush;
        }
        std::fflush(stdout);
        std::_Exit(99);
    }
    void finish_writer(std::future<int> &w)
    {
        if (w.valid() && w.get()) shared->exit_code = 1;
    }

    // A scene file parsed (scene.h:212-452 -> rrtx_scene_load), with what the reference would have exited with had it failed
    struct Parsed {
        rrtx_scene *scene = nullptr;
        int rc = 0, exit_code = 0;
        std::string message; // the parser's diagnostic: printed with the job's own block of stderr lines, not by the helper thread
    };
    Parsed parse(const Job &job) const
    {
        Parsed p;
        p.rc = rrtx_scene_load_quiet(job.scene_file.c_str(), prm.image_width, prm.image_height, kFp64, &p.scene);
        if (p.rc) p.exit_code = rrtx_scene_exit_code(), p.message = rrtx_scene_error(); // (thread-local: read on the thread that parsed)
        return p;
    }

    void run()
    {
        // the NEXT scene of this worker is parsed by a helper thread while the current one renders (0.4 ms of a 5.2 ms frame at
        // the reference's animation settings): a worker therefore holds one job in reserve
        const std::vector<Job> &jobs = *shared->jobs;
        size_t cur = shared->next_job.fetch_add(1);
        Parsed cur_scene;
        if (cur < jobs.size()) cur_scene = parse(jobs[cur]);
        while (cur < jobs.size()) {
            const size_t nxt = shared->stop ? jobs.size() : shared->next_job.fetch_add(1);
            std::future<Parsed> ahead;
            if (nxt < jobs.size()) ahead = std::async(std::launch::async, [this, &jobs, nxt]() { return parse(jobs[nxt]); });
            if (shared->stop) { // a scene elsewhere did not parse: nothing further is started (what is in flight is written)
                skip_output_turn(jobs[cur]);
                if (cur_scene.scene) rrtx_scene_free(cur_scene.scene);
            }
            else
                render_job(jobs[cur], cur_scene);
            cur = nxt;
            cur_scene = ahead.valid() ? ahead.get() : Parsed();
        }
        for (auto &w : writers) finish_writer(w);
        for (auto &f : frames)
            if (f) (void)rrtx_unpin_host(f->data());
        if (ctx) rrtx_destroy(ctx);
        if (group) rrtx_group_destroy(group);
    }

    void skip_output_turn(const Job &job)
    {
        // a job that will not print its PPM must not hold up the ones behind it
        if (job.ppm_index == (size_t)-1) return;
        std::unique_lock<std::mutex> lock(shared->out_mu);
        shared->out_cv.wait(lock, [&] { return shared->next_ppm == job.ppm_index; });
        shared->next_ppm += 1;
        shared->out_cv.notify_all();
    }

    void render_job(const Job &job, const Parsed &parsed)
    {
        std::ostringstream err; // this job's stderr chatter
        rrtx_scene *scene = parsed.scene;
        int rc = parsed.rc;
        if (rc) {
            // the reference exits on the spot with scene.h's code (1 obj errors, 2 cannot open, 3 unknown material, 4 sanity);
            // in a batch: no further scenes are started, what is in flight is written, the first such code is the exit code
            int code = parsed.exit_code;
            if (!parsed.message.empty()) log(parsed.message + "\n");
            int expected = 0;
            shared->exit_code.compare_exchange_strong(expected, code ? code : 1);
            shared->stop = true;
            skip_output_turn(job);
            return;
        }
        int32_t counts[6];
        rrtx_scene_counts(scene, counts);
        rrtx_scene_desc desc;
        rrtx_scene_describe(scene, &desc);
        // scene.h:443-451
        err << "read scene file: " << job.scene_file << "\n";
        err << "material count:  " << counts[0] << "\n";
        err << "sphere count:    " << counts[1] << "\n";
        err << "msphere count:   " << counts[2] << "\n";
        err << "obj count:       " << counts[4] << "\n";
        err << "obj_inst count:  " << counts[5] << "\n";
        {
            const fp_t *cam = (const fp_t *)desc.camera;
            if (cam[22] != cam[23]) err << "camera time:     " << cam[22] << " - " << cam[23] << "\n";
        }

        std::time_t render_time = std::time(nullptr);
        std::tm render_tm;
        localtime_r(&render_time, &render_tm);

        if (group_size > 1) {
            if (!group) {
                // RCCL greets on stdout while the communicators are built, and stdout is where the PPM goes (main.cpp:142): file descriptor 1
                // points at stderr for the duration.  This is the host's business, not the library's, and it is safe HERE: a group is created
                // by the first job of this worker, before any writer task exists.
                fflush(stdout);
                const int saved_stdout = ::dup(1);
                if (saved_stdout >= 0) (void)::dup2(2, 1);
                rc = rrtx_group_create(&prm, group_size, group_devices.data(), group_rehearsal ? RRTX_GROUP_REHEARSAL : 0, &group);
                fflush(stdout);
                if (saved_stdout >= 0) {
                    (void)::dup2(saved_stdout, 1);
                    ::close(saved_stdout);
                }
                if (rc) die(rc, err.str());
            }
            rc = rrtx_group_set_scene(group, &desc);
        }
        else {
            if (!ctx) {
                rc = rrtx_create(&prm, &ctx);
                if (rc) die(rc, err.str());
            }
            rc = rrtx_set_scene(ctx, &desc);
        }
        if (rc) die(rc, err.str());

        // frame buffers used in turn: while this scene renders into one, writer tasks quantise and encode the previous ones
        // (allocating and zero-filling 11 MB per frame cost more than a short render)
        const size_t turn = rendered;
        rendered += 1;
        // (the buffer of this turn was last read by the writer of turn - (kWriters + 1), whose slot, (turn - 1) % kWriters, was
        // joined when the writer of turn - 1 took it over: nothing to wait for here)
        std::shared_ptr<std::vector<fp_t>> &slot = frames[turn % (kWriters + 1)];
        if (!slot) {
            slot = std::make_shared<std::vector<fp_t>>((size_t)prm.image_width * prm.image_height * 3, (fp_t)0);
            (void)rrtx_pin_host(slot->data(), slot->size() * sizeof(fp_t)); // (best effort: unpinned it is only slower)
        }
        auto fb = slot;
        // rrt.cu:195-202,261
        err << "HIP Runtime Version " << rrtx_runtime_version() << "\n";
        err << "Rendering a " << prm.image_width << "x" << prm.image_height << " image with " << prm.samples_per_pixel << " samples per pixel ";
        err << "(" << kFpName << (prm.use_bvh ? ", acceleration grid where the scene allows; same bits as -b for spheres and fp64 meshes" : ", list scan") << ").\n";
        err << "num_hittables = " << (counts[1] + counts[2] + counts[3]) << "\n";

        double seconds = 0, samples = 0, bytes_algorithmic = 0;
        unsigned long long segments = 0, prim_tests = 0;
        int blocks = 0, accel_cells = 0, accel_exact = 1, scan_mfma = 0;
        std::string how;
        if (group_size > 1) {
            err << "HIP Devices:";
            for (int d : group_devices) err << " " << d;
            err << (group_rehearsal ? " (rehearsal: members share devices, no RCCL)" : " (row tiles of " + std::to_string(prm.tile_rows > 0 ? prm.tile_rows : 4) + ", RCCL gather to the first)") << "\n";
            rrtx_group_stats gs;
            std::memset(&gs, 0, sizeof gs);
            rc = rrtx_group_render(group, fb->data(), &gs);
            if (rc) die(rc, err.str());
            seconds = gs.device_ms / 1000.0, samples = (double)gs.samples, bytes_algorithmic = (double)gs.bytes_algorithmic;
            segments = gs.segments, prim_tests = gs.prim_tests, accel_cells = gs.accel_cells, accel_exact = gs.accel_exact;
            std::ostringstream h;
            h << group_size << " devices: slowest render " << gs.render_ms / 1000.0 << " s, gather " << gs.gather_ms / 1000.0 << " s (" << gs.gathered_bytes << " bytes" << (gs.rccl ? ", RCCL" : ", copies") << "),";
            how = h.str();
        }
        else {
            err << "HIP Device: " << prm.device << "\n";
            rrtx_stats st;
            std::memset(&st, 0, sizeof st);
            rc = rrtx_render(ctx, fb->data(), &st);
            if (rc) die(rc, err.str());
            seconds = st.kernel_ms / 1000.0, samples = (double)st.samples, bytes_algorithmic = (double)st.bytes_algorithmic;
            segments = st.segments, prim_tests = st.prim_tests, blocks = st.grid_blocks, accel_cells = st.accel_cells, accel_exact = st.accel_exact, scan_mfma = st.scan_mfma;
        }
        err << "took " << seconds << " seconds.\n";
        char hostname[HOST_NAME_MAX + 1];
        hostname[0] = 0;
        gethostname(hostname, sizeof hostname);
        char when[128];
        std::strftime(when, sizeof when, "%c %Z,", &render_tm);
        // rrt.cu:312-315: stats,<date>,<host>,<runtime>,<fp>,w,h,spp,blocks,tx,ty,seconds
        err << "stats," << when << hostname << ",HIP" << rrtx_runtime_version() << "," << kFpName << "," << prm.image_width << "," << prm.image_height << "," << prm.samples_per_pixel << ","
            << blocks << "," << prm.threads_x << "," << prm.threads_y << "," << seconds << "\n";
        if (seconds > 0)
            err << "rate," << samples / seconds / 1e6 << " Msamples/s," << how << segments << " segments," << prim_tests << " primitive tests," << bytes_algorithmic / seconds / 1e9
                << " GB/s algorithmic," << (accel_cells ? "grid of " + std::to_string(accel_cells) + " cells" : std::string(scan_mfma ? "list scan (filter on the matrix cores)" : "list scan"))
                << (accel_exact ? "" : ",approximate rule (fp32 triangles gridded under an empirical inflation: -X or -b for the list scan's bits)") << "\n";
        rrtx_scene_free(scene);
        log(err.str());

        // main.cpp:140-162: quantise, flip, write - off the render path: it overlaps this worker's next scene
        std::future<int> &writer = writers[turn % kWriters];
        finish_writer(writer); // (the task of kWriters turns ago)
        const int w = prm.image_width, h = prm.image_height, spp = prm.samples_per_pixel;
        Shared *sh = shared;
        const Job *jp = &job;
        writer = std::async(std::launch::async, [fb, jp, sh, w, h, spp]() -> int {
            std::vector<uint8_t> rgb((size_t)w * h * 3);
            rrtx_quantise(fb->data(), kFp64, w, h, spp, rgb.data());
            int wrc;
            if (jp->png_file)
                wrc = rrtx_write_png(jp->png_file, rgb.data(), w, h);
            else {
                // stdout is one stream: the PPMs of a batch come out in job order, whoever rendered them
                std::unique_lock<std::mutex> lock(sh->out_mu);
                sh->out_cv.wait(lock, [&] { return sh->next_ppm == jp->ppm_index; });
                wrc = rrtx_write_ppm(nullptr, rgb.data(), w, h);
                sh->next_ppm += 1;
                sh->out_cv.notify_all();
            }
            if (wrc) {
                std::lock_guard<std::mutex> lock(sh->log_mu);
                std::cerr << "ERROR: could not write image\n";
            }
            return wrc;
        });
    }
};

int main(int argc, char *argv[])
{
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); // (the hosts of this pool only support dmabuf IPC; RCCL wants it set before the runtime starts)
    if (rrtx_abi_version() != RRTX_ABI_VERSION) { // the library fills this binary's rrtx_stats / rrtx_group_stats with ITS sizeof
        std::cerr << "librrtx.so has struct layout version " << rrtx_abi_version() << ", this binary was built for " << RRTX_ABI_VERSION << "\n";
        return 99;
    }
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
    int n_gpus = 1;
    bool rehearse = false;

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
        case 'G': n_gpus = atoi(next()); break;
        case 'E': rehearse = true; break;
        case 'X': prm.flags |= RRTX_FLAG_EXACT_ACCEL; break;
        default: usage(argv[i]);
        }
    }

    if (scene_files.empty()) {
        std::cerr << "ERROR: no scene loaded." << std::endl;
        return 1;
    }

    // ---- who renders what --------------------------------------------------------------------------
    // -G 1 (default): one device context, the scenes one after another.
    // -G n, fewer scenes than GPUs: every frame is cut into row tiles over the n GPUs (rrtx_group: one RCCL gather).
    // -G n, at least n scenes: the scenes are dealt to n workers, one device each - frame-level parallelism, no
    //       communication at all (the reference's 261-frame animation job, scenes/final_anim/Makefile).
    const int available = rrtx_device_count();
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > 1 && !rehearse && n_gpus > available) {
        std::cerr << "HIP error = -1 : -G " << n_gpus << " but only " << available << " device(s) present\n";
        return 99;
    }
    std::vector<Job> jobs;
    size_t n_ppm = 0;
    for (size_t k = 0; k < scene_files.size(); ++k) {
        Job j;
        j.scene_file = scene_files[k], j.png_file = k < png_files.size() ? png_files[k] : nullptr;
        j.ppm_index = j.png_file ? (size_t)-1 : n_ppm++;
        jobs.push_back(j);
    }
    auto device_of = [&](int k) { return available > 0 ? (prm.device + k) % available : prm.device + k; };
    const bool frame_level = n_gpus > 1 && jobs.size() >= (size_t)n_gpus;
    Shared shared;
    shared.jobs = &jobs;
    std::vector<Worker> workers(frame_level ? n_gpus : 1);
    for (size_t k = 0; k < workers.size(); ++k) {
        Worker &w = workers[k];
        w.prm = prm;
        w.shared = &shared;
        if (frame_level)
            w.prm.device = device_of((int)k);
        else if (n_gpus > 1) {
            w.group_size = n_gpus, w.group_rehearsal = rehearse;
            for (int q = 0; q < n_gpus; ++q) w.group_devices.push_back(device_of(q));
        }
    }
    if (workers.size() == 1)
        workers[0].run();
    else {
        std::vector<std::thread> threads;
        for (Worker &w : workers) threads.emplace_back([&w]() { w.run(); });
        for (std::thread &t : threads) t.join();
    }
    return shared.exit_code.load();
}
