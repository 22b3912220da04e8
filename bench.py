#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the render hot path on scenes/final.txt (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over the workload: render the frame (scene tables already resident in HBM,
framebuffer left in HBM) and, for N > 1, the one gather of the row-tile shards to rank 0 over RCCL.

Workloads (BASELINE.json `configs`):
  N = 1   C3 = scenes/final.txt 1200x800 spp 500 d 50 fp32, brute-force list scan (the mode north_star names and
          the roofline is defined for) - the configuration the metric is quoted on.  The same line carries, timed by
          this run: `accelerated` (C3 with use_bvh, the CLI's default mode) and `configs` = C2 (test1 1200x800
          spp 10), C4 (final fp64), C5 on ONE GPU (final 3840x2160 spp 1000), each in both modes, and `mesh` (a 27 072-
          triangle mesh).
  N > 1   C5 = scenes/final.txt 3840x2160 spp 1000 fp32 cut into row tiles over the N ranks ("scaling": "strong":
          the total is BASELINE.json's 8-GPU job whatever N is), one RCCL gather to rank 0 per step; the line reports
          the kernel-only time (slowest rank) next to the gather-inclusive step time, `config.rccl` says what RCCL saw
          (ranks, backend, version, every rank's device), `one_gpu_same_workload` times the whole C5 frame on rank 0's GPU
          alone inside the job, and `strong_c3` splits configuration 3 the same way (north_star states its 7.5 x on C3:
          9.7 ms of work per GPU at N = 8) next to C3 whole on rank 0's GPU.  `--strong-c3` makes C3 the line's own
          workload, `--weak` keeps 480 M samples per GPU (C3 at spp 500 x N).
The image is bit-identical for every N and tile size (tests/test_gpu_configs.py, tests/test_gpu_group.py).

Rank 0 prints ONE JSON line.  Besides the contract's fields:
  roofline     - for the dominant kernel, against the unit that BINDS it.  The scene is 7.8 KB (no HBM stream); the scan's
                 filter is a dot product of 31 f16 terms per (ray, sphere) and runs on the matrix cores (two chained
                 v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays), everything else - camera rays, exact tests, shading - on
                 the vector unit, and both are issued through the SIMD's one vector port: `achieved` = wave-instructions per
                 clock per SIMD with a matrix instruction counted as the 4 it keeps out (8 clocks), `peak` = 0.5 (one
                 wave-instruction every second clock), `frac` = achieved / peak (a model where matrix instructions are weighed, clamped to 1; the measured rates are beside it).  `roofline.mfma` prices the matrix
                 instructions alone against the dense f16 peak (2.5 PFLOP/s): the filter is a third of the kernel's cycles.
                 At N = 1 the counters come from rocprofv3 --pmc passes THIS run makes (SQ_INSTS_VALU & co. in one
                 pass, the matrix pipe's in another, FETCH_SIZE and WRITE_SIZE in passes of their own - the HBM `traffic`), on the `rrt` binary
                 rendering the same workload through the same library, before this process touches the GPU.
                 `logical_hbm` keeps SURVEY.md 8(d)'s figure - algorithmic bytes of the reference's list scan / kernel
                 time against 8 TB/s - as a named field: it exceeds 1 by construction (one 16-byte record read serves
                 64 rays, camera rays are resolved from per-pixel lists) and says nothing about kernel quality.
  cpu_baseline - the reference's own OpenMP binary (oracle/_ref/rrto, built from the reference sources) timed here
                 on the host cores, on a bounded sample of the same scene.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE = os.path.join(ROOT, "scenes", "final.txt")
TEST1 = os.path.join(ROOT, "scenes", "test1.txt")
DEPTH = 50
C3 = (1200, 800, 500)
C5 = (3840, 2160, 1000)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK = 0.5        # VALU wave-instructions per clock per SIMD: a wave64 instruction occupies the 32-lane fp32 pipe for 2 clocks
MFMA_ISSUE = 4         # ... and a matrix instruction holds that issue port for 8 clocks (MI355X_MICROARCH.md, cycle constants): 4 plain instructions' worth
MFMA_PEAK_TFLOPS = 2500.0  # dense f16 (MI355X_MICROARCH.md)
MFMA_FLOP = 2 * 32 * 32 * 16  # per v_mfma_f32_32x32x16_f16
MF_BLOCK = 32          # spheres per block of the filter's table (rrtx_pack.h: kMfBlock)
N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9       # nominal; a PMC pass measures the real one (GRBM_GUI_ACTIVE / 8 XCDs / duration)
# Names as rocprofv3 prints them, up to and including the RESUME = false argument - prefixes: behind it stands SOV, the variant a launch selects by what it knows
# about its scene and itself (spheres alone, a first-bounce pre-pass, no single-sample tasks: rrtx_kernels.hip); ONE of them runs per launch
# (tests/test_cabi.py checks the names against the library's symbols): the list-scan render kernel of final.txt and the accelerated one.
LIST_KERNEL = "render_kernel<float, true, 3, false, 0, false"
ACCEL_KERNEL = "render_kernel<float, true, 0, false, 2, false"
# the mesh sub-result: tables in HBM (ACCEL = 1), scenes of every kind (SOV = 0), the render pass and the resume pass (RESUME = true) that finishes what it parks
MESH_RENDER_KERNEL = "render_kernel<float, true, 0, false, 1, false, 0>"
MESH_RESUME_KERNEL = "render_kernel<float, true, 0, false, 1, true, 0>"
MESH = (48, 96, 600, 400, 16)  # UV sphere of 48 x 96 quads instanced three times = 27 072 triangles; frame and spp of tests/test_gpu_mesh.py and VERDICT r02


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (the reference's own binaries; test infrastructure, never the thing measured as `value`)
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline():
    """Reference CPU build on a bounded sample (~10-30 s of CPU work)."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    rrto, rrtc = os.path.join(ref, "rrto"), os.path.join(ref, "rrtc")

    share = max(1, min(os.cpu_count() or 1, 16))  # one GPU's share of the host (16 cores on the MI355X boxes)
    omp_env = dict(os.environ, OMP_NUM_THREADS=str(share))

    def run(exe, args, env=None):
        t = time.time()
        p = subprocess.run([exe, "-i", SCENE] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=600)
        wall = time.time() - t
        err = p.stderr.decode(errors="replace")
        m = re.search(r"took ([0-9.eE+-]+) seconds", err)
        st = re.search(r"^stats,.*$", err, flags=re.M)
        threads = None
        if st:
            f = st.group(0).split(",")
            threads = f[-4]
        return (float(m.group(1)) if m else wall), threads

    if os.path.exists(rrto) and os.path.exists(rrtc):
        w, h, s = 1200, 800, 10
        sec, threads = run(rrto, ["-w", str(w), "-h", str(h), "-s", str(s)], env=omp_env)  # the reference's default (CPU BVH on)
        out = {"value": round(w * h * s / sec / 1e6, 4), "unit": "Msamples/s", "cores": int(threads) if threads and threads.isdigit() else share, "kind": "reference",
               "sample": "oracle/_ref/rrto (reference OpenMP fp64 build, its default CPU BVH) on scenes/final.txt %dx%d spp=%d d=50: %.2f s" % (w, h, s, sec)}
        w2, h2, s2 = 600, 400, 4
        sec_b, _ = run(rrto, ["-w", str(w2), "-h", str(h2), "-s", str(s2), "-b"], env=omp_env)
        out["brute_force_value"] = round(w2 * h2 * s2 / sec_b / 1e6, 4)
        out["brute_force_sample"] = "rrto -b (list scan, the mode the headline HIP kernel implements) %dx%d spp=%d: %.2f s" % (w2, h2, s2, sec_b)
        sec_c, _ = run(rrtc, ["-w", str(w2), "-h", str(h2), "-s", str(s2)])
        out["single_thread_value"] = round(w2 * h2 * s2 / sec_c / 1e6, 4)
        out["single_thread_sample"] = "oracle/_ref/rrtc (1 thread fp32, BVH) %dx%d spp=%d: %.2f s" % (w2, h2, s2, sec_c)
        return out
    # no reference build on this box: time the oracle (CPU port), all cores
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle

    w, h, s = 600, 400, 8
    o = Oracle(SCENE, w, h, False)
    t = time.time()
    o.render(s, DEPTH, 1984, order=1)
    sec = time.time() - t
    return {"value": round(w * h * s / sec / 1e6, 4), "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "oracle/librrt_oracle.so (OpenMP port, brute-force list scan, fp32) on scenes/final.txt %dx%d spp=%d: %.2f s" % (w, h, s, sec)}


# ------------------------------------------------------------------------------------------------------------------
# live PMC passes: rocprofv3 on the `rrt` binary (a separate process), BEFORE this process initialises the GPU
# ------------------------------------------------------------------------------------------------------------------
def kernel_source_hash():
    """What the committed fallback profile is checked against: the device code it was measured on."""
    h = hashlib.sha256()
    for f in ("rrtx_kernels.hip", "rrtx_path.h", "rrtx_device.h", "rrtx_grid.h"):
        h.update(open(os.path.join(ROOT, "rrt_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def mesh_scene_file():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mesh_gen import mesh_scene

    d = tempfile.mkdtemp(prefix="rrtx_mesh_", dir="/tmp")
    return mesh_scene(os.path.join(d, "mesh.txt"), MESH[0], MESH[1])


def pmc_pass(counters, rrt_args, want_kernel, timeout=100):
    """One `rocprofv3 --pmc <counters>` run (no tracing) of ./rrt <rrt_args>; -> {counter: per-launch value of
    `want_kernel`, "duration_ms": ...} or None.  The program itself follows `--` (no env / shell hop)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="rrtx_pmc_", dir="/tmp")
    try:
        png = os.path.join(out, "frame.png")
        cmd = [exe, "--output-format", "csv", "--pmc"] + list(counters) + ["-d", out, "-o", "p", "--", os.path.join(ROOT, "rrt")] + list(rrt_args) + ["-o", png]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
        if r.returncode != 0:
            return None
        acc, n, dur, name = {}, {}, [], None
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if want_kernel in row["Kernel_Name"]:
                    name = row["Kernel_Name"]
                    c = row["Counter_Name"]
                    acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
                    n[c] = n.get(c, 0) + 1
                    try:
                        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
                    except Exception:
                        pass
        if not acc:
            return None
        res = {c: acc[c] / n[c] for c in acc}  # one row per (dispatch, counter): the mean over this kernel's dispatches = per launch
        res["launches"] = max(n.values())
        res["kernel_name"] = name
        if dur:
            res["duration_ms"] = sum(dur) / len(dur)
        return res
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_pmc(w, h, spp):
    """The passes behind roofline.frac and roofline.traffic (list scan) and the accelerated kernel's lane utilisation."""
    base = ["-i", SCENE, "-w", str(w), "-h", str(h), "-s", str(spp), "-d", str(DEPTH)]
    sq = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    mf = ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16", "GRBM_GUI_ACTIVE"]  # the matrix pipe, a pass of its own
    res = {"source": "rocprofv3 --pmc passes made by this bench.py run on `rrt` (same library, same workload), one counter group per pass, no tracing",
           "kernel_source_sha": kernel_source_hash()}
    a = pmc_pass(sq, base + ["-b"], LIST_KERNEL)
    if not a:
        return None
    res["list_scan"] = a
    m = pmc_pass(mf, base + ["-b"], LIST_KERNEL)
    if m:
        res["list_scan_mfma"] = {k: v for k, v in m.items() if k != "kernel_name"}
    f = pmc_pass(["FETCH_SIZE"], base + ["-b"], LIST_KERNEL)
    wv = pmc_pass(["WRITE_SIZE"], base + ["-b"], LIST_KERNEL)
    if f and wv:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KB) reports half of the bytes read on gfx950; WRITE_SIZE (KB) is exact
        res["hbm_read_bytes"] = int(f["FETCH_SIZE"] * 1024 * 2)
        res["hbm_write_bytes"] = int(wv["WRITE_SIZE"] * 1024)
    g = pmc_pass(sq, base, ACCEL_KERNEL)
    if g:
        res["accelerated"] = g
    # the mesh scene (tables in HBM): HBM traffic and VALU counters of its two passes
    try:
        mesh_file, _ = mesh_scene_file()
        margs = ["-i", mesh_file, "-w", str(MESH[2]), "-h", str(MESH[3]), "-s", str(MESH[4]), "-d", str(DEPTH)]
        mesh = {}
        for name, kern in (("render", MESH_RENDER_KERNEL), ("resume", MESH_RESUME_KERNEL)):
            q = pmc_pass(sq, margs, kern)
            fr, wr = pmc_pass(["FETCH_SIZE"], margs, kern), pmc_pass(["WRITE_SIZE"], margs, kern)
            if q and fr and wr:
                q["hbm_read_bytes"], q["hbm_write_bytes"] = int(fr["FETCH_SIZE"] * 1024 * 2), int(wr["WRITE_SIZE"] * 1024)
                mesh[name] = q
        if mesh:
            res["mesh"] = mesh
    except Exception:
        pass
    return res


def committed_pmc():
    """Fallback when no pass can be made here: the committed profile, only if it was measured on THIS device code."""
    p = os.path.join(ROOT, "profiles", "r04_pmc_live.json")
    try:
        d = json.load(open(p))
    except Exception:
        return None
    if d.get("kernel_source_sha") != kernel_source_hash():
        return None  # stale: the kernels changed since it was taken
    d["source"] = "profiles/r04_pmc_live.json (committed; measured on the same device code: sha %s)" % d["kernel_source_sha"]
    return d


def valu_numbers(pm):
    """-> (wave-instructions / clk / SIMD, lane utilisation, clock GHz or None) from one SQ pass."""
    cycles = pm["GRBM_GUI_ACTIVE"] / 8.0  # per XCD
    per_clk = pm["SQ_INSTS_VALU"] / (N_SIMD * cycles)
    lanes = pm["SQ_THREAD_CYCLES_VALU"] / (64.0 * pm["SQ_ACTIVE_INST_VALU"]) if pm.get("SQ_ACTIVE_INST_VALU") else None
    ghz = cycles / (pm["duration_ms"] * 1e-3) / 1e9 if pm.get("duration_ms") else None
    return per_clk, lanes, ghz


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (non-headline runs only)")
    ap.add_argument("--strong-c3", action="store_true", help="N > 1: shard configuration 3 (1200x800 spp 500) instead of configuration 5")
    ap.add_argument("--weak", action="store_true", help="N > 1: keep the work per GPU fixed (configuration 3 at spp 500 x N)")
    ap.add_argument("--tile-rows", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accel", action="store_true", help="skip the extra use_bvh measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C4 / C5-on-one-GPU sub-results")
    ap.add_argument("--no-one-gpu", action="store_true", help="N > 1: skip timing the same workload on rank 0 alone")
    ap.add_argument("--no-strong-c3", action="store_true", help="N > 1: skip the extra split of configuration 3 (the `strong_c3` sub-result)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 passes (a committed profile is used if it matches the device code)")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: RCCL greets on stdout ("RCCL version : ...") when a communicator is
    # built, child processes may chatter - so file descriptor 1 points at stderr for the duration of the run and the
    # line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything initialises the HIP runtime (RCCL IPC on this pool)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))

    # workload
    if world == 1 or args.strong_c3 or args.weak:
        W, H, spp = C3
        if args.weak:
            spp *= world
        wl_name = "C3"
    else:
        W, H, spp = C5
        wl_name = "C5"
    if args.spp > 0:
        spp = args.spp
    headline = world == 1 and (W, H, spp) == C3

    # ---- PMC first: this process has not touched the GPU yet, rocprofv3 profiles a child (`rrt`)
    pmc = None
    # (never from inside a profiler: `rocprofv3 -- python3 bench.py` must not start a second one underneath itself)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and headline and not args.no_pmc and not under_profiler:
        try:
            pmc = live_pmc(W, H, spp)
        except Exception:
            pmc = None
    if world == 1 and headline and pmc is None:
        pmc = committed_pmc()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible - the render path has no CPU fallback")
    # one rank per GPU.  RRTX_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks (ranks then share devices and the gather is staged through the host): it checks the
    # plumbing, its numbers mean nothing.
    backend = os.environ.get("RRTX_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    rccl_info = None
    if world > 1:
        # a communicator that cannot be built ends the run, non-zero, with RCCL's own words on stderr: no re-exec, no silent
        # fall-back to another backend (RRTX_BENCH_BACKEND=gloo is something the caller asks for, see above)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
                probe = torch.ones(1, device=torch.device("cuda", device_index))
                dist.all_reduce(probe)  # builds the communicator now: an RCCL failure surfaces here, with its text
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            else:
                dist.init_process_group(backend)
        except Exception as e:
            sys.stderr.write("bench.py: rank %d: the %s process group could not be built: %r\n" % (rank, backend, e))
            sys.stderr.flush()
            os._exit(3)
        # what RCCL saw, for the reader of the line: ranks, backend, every rank's device
        mine = {"rank": rank, "local_rank": local_rank, "ordinal": device_index, "name": torch.cuda.get_device_name(device_index), "pid": os.getpid(), "host": os.uname().nodename}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ver = None
        rccl_info = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ver if backend == "nccl" else None, "devices": everyone,
                     "distinct_devices": len({(d["host"], d["ordinal"]) for d in everyone})}
    dev = torch.device("cuda", device_index)

    import rrt_amd
    from rrt_amd.dist import ShardedRenderer

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if world > 1:
            tt = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return x

    def timed(sr, steps, warmup, gather=True):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks.  -> (seconds, stats)"""
        step = sr.render if gather else sr.render_local
        for _ in range(warmup):
            step()
        barrier()
        sr.rrt.collect()  # drop warm-up launches from the event statistics
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        return reduce_max(time.perf_counter() - t0), sr.rrt.collect()

    sr = ShardedRenderer(SCENE, W, H, spp, DEPTH, fp64=False, tile_rows=args.tile_rows, device=dev, collect_stats=True)
    if world > 1:
        # RCCL sets its point-to-point channels up at their first use: one untimed exchange of a few bytes, so that not
        # even a run with --warmup 0 times connection set-up instead of the gather
        tiny = torch.zeros(8, dtype=torch.float32, device="cuda" if backend == "nccl" else "cpu")
        dist.gather(tiny, gather_list=[torch.zeros_like(tiny) for _ in range(world)] if rank == 0 else None, dst=0)
        barrier()
    elapsed, st = timed(sr, args.steps, args.warmup)
    kernel_only = None
    if world > 1:  # the same steps without the gather: what the exchange adds
        ko, _ = timed(sr, args.steps, 0, gather=False)
        kernel_only = ko / args.steps * 1e3

    # the same steps with use_bvh (the CLI's default): reported beside the headline, which
    # stays the list scan the north star names and the roofline is defined for
    accel = None
    if not args.no_accel:
        sa = ShardedRenderer(SCENE, W, H, spp, DEPTH, fp64=False, tile_rows=args.tile_rows, device=dev, collect_stats=True, use_bvh=True)
        elapsed_a, sta = timed(sa, args.steps, 1)
        accel = {"value": round(W * H * spp / (elapsed_a / args.steps) / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(elapsed_a / args.steps * 1e3, 3),
                 "kernel_ms": round(sta["kernel_ms_sum"] / max(1, sta["renders"]), 3), "grid_cells": sta["accel_cells"],
                 "sky_pixels": int(sta.get("sky_pixels", 0)), "first_bounce_prepass": int(sta.get("first_bounce", 0)),
                 "note": "use_bvh = 1 (the CLI's default, as the reference's BVH is): closest hit through a uniform grid + always-list, exact test and tie rules of the list scan, "
                         "image bit-identical for spheres, moving spheres and fp64 meshes - this scene: yes (tests/test_gpu_configs.py; fp32 triangle meshes are gridded under a stated "
                         "tolerance, rrtx_stats.accel_exact = 0, `rrt -X` / -b for the list scan's bits); segments fall back to the list scan only for rays outside the grid's proven range (DESIGN.md 3b)"}
        if pmc and pmc.get("accelerated"):
            pc, lanes, _ = valu_numbers(pmc["accelerated"])
            accel["valu_issue_frac"] = round(pc / VALU_PEAK, 4)
            accel["valu_lane_utilisation"] = round(lanes, 4) if lanes else None
        del sa

    # ---- the other configurations of BASELINE.json, timed by this run (N = 1 only)
    configs = None
    if world == 1 and headline and not args.no_configs:
        configs = {}

        def one(name, scene, w, h, s, fp64, steps):
            ent = {"workload": "%s %dx%d spp=%d d=%d %s on 1 GPU" % (os.path.relpath(scene, ROOT), w, h, s, DEPTH, "fp64" if fp64 else "fp32")}
            for mode, bvh in (("list_scan", False), ("use_bvh", True)):
                r = ShardedRenderer(scene, w, h, s, DEPTH, fp64=fp64, tile_rows=args.tile_rows, device=dev, collect_stats=True, use_bvh=bvh)
                sec, stt = timed(r, steps, 1)
                ent[mode] = {"ms_per_step": round(sec / steps * 1e3, 3), "kernel_ms": round(stt["kernel_ms_sum"] / max(1, stt["renders"]), 3), "Msamples_per_s": round(w * h * s / (sec / steps) / 1e6, 1),
                             "sample_chunk": stt["sample_chunk"], "grid_cells": stt["accel_cells"]}
                del r
            configs[name] = ent

        # a triangle mesh, its tables (1.4 MB of triangles, 1.3 MB of cell headers, 0.4 MB of cell lists) read from HBM / L2
        try:
            mesh_file, n_tri = mesh_scene_file()
            mw, mh, ms = MESH[2], MESH[3], MESH[4]
            ent = {"workload": "UV-sphere mesh x 3 instances = %d triangles + 101 spheres (tools/mesh_gen.py), %dx%d spp=%d d=%d fp32, use_bvh on 1 GPU" % (n_tri, mw, mh, ms, DEPTH)}
            rm = ShardedRenderer(mesh_file, mw, mh, ms, DEPTH, fp64=False, tile_rows=args.tile_rows, device=dev, collect_stats=True, use_bvh=True)
            secm, stm = timed(rm, 20, 2)
            kms = stm["kernel_ms_sum"] / max(1, stm["renders"])
            # algorithmic bytes of the walk: 8 B of list header per visited cell + (4 B index + 36 B of vertices / 16 B of sphere) per tested pair
            logical = stm["walk_cells"] * 8 + stm["walk_pairs"] * 40
            ent.update({"ms_per_step": round(secm / 20 * 1e3, 3), "kernel_ms": round(kms, 3), "Msamples_per_s": round(mw * mh * ms / (secm / 20) / 1e6, 1), "segments": stm["segments"], "grid_cells": stm["accel_cells"],
                        "accel_exact": stm["accel_exact"], "cells_visited": stm["walk_cells"], "pairs_tested": stm["walk_pairs"],
                        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes": int(logical), "achieved": round(logical / (kms * 1e-3) / 1e9, 1),
                                     "frac": round(logical / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "note": "algorithmic bytes = cells visited x 8 B (list header) + (ray, entry) pairs tested x 40 B (index + the 9 numbers of the test), over the kernel time of the whole launch "
                                             "(render pass + resume pass); accel_exact = 0: fp32 triangles are gridded under the approximate rule (include/rrtx.h)"}})
            if pmc and pmc.get("mesh"):
                tr = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in pmc["mesh"].values())
                ent["roofline"]["traffic"] = int(tr)
                ent["roofline"]["traffic_GBs"] = round(tr / (kms * 1e-3) / 1e9, 1)
                ent["passes"] = {k: {"kernel_ms": round(v.get("duration_ms", 0.0), 3), "valu_issue_frac": round(valu_numbers(v)[0] / VALU_PEAK, 4), "valu_lane_utilisation": round(valu_numbers(v)[1] or 0.0, 4),
                                     "hbm_read_bytes": v["hbm_read_bytes"], "hbm_write_bytes": v["hbm_write_bytes"]} for k, v in pmc["mesh"].items()}
            configs["mesh"] = ent
            del rm
        except Exception as e:
            configs["mesh"] = {"error": repr(e)}
        one("C2", TEST1, 1200, 800, 10, False, 20)
        one("C4", SCENE, 1200, 800, 500, True, 3)
        one("C5_on_1_gpu", SCENE, C5[0], C5[1], C5[2], False, 2)
        # ... and the native single-process group (rrtx_group: RCCL send / recv + de-interleave) with the devices of this box
        try:
            g = rrt_amd.RrtGroup(torch.cuda.device_count(), W, H, spp, DEPTH, use_bvh=False, tile_rows=args.tile_rows)
            g.set_scene(rrt_amd.Scene(SCENE, W, H))
            g.render_device()
            ds = []
            for _ in range(3):
                g.render_device()
                ds.append(dict(g.stats))
            best = min(ds, key=lambda d: d["device_ms"])
            configs["C3_native_group"] = {"n_devices": best["n_devices"], "rccl": best["rccl"], "render_ms": round(best["render_ms"], 3), "device_ms": round(best["device_ms"], 3),
                                          "gather_ms": round(best["gather_ms"], 3), "gathered_bytes": best["gathered_bytes"], "rccl_comms": best["rccl_comms"], "rccl_version": best["rccl_version"],
                                          "devices": best["devices"],
                                          "note": "rrtx_group_render_device: one process, ncclCommInitAll + grouped ncclSend / ncclRecv to device 0 + de-interleave; device_ms = first launch -> assembled frame"}
            g.close()
        except Exception as e:
            configs["C3_native_group"] = {"error": repr(e)}

    # ---- N > 1: the SAME workload on one GPU, timed by rank 0 inside this job (the others wait): what `value` is N times of
    # when the split is perfect.  (The N = 1 bench line is configuration 3; the N > 1 default is configuration 5.)
    one_gpu = None
    if world > 1 and not args.no_one_gpu:
        barrier()
        if rank == 0:
            try:
                r1 = rrt_amd.Rrt(W, H, spp, DEPTH, use_bvh=False, fp64=False, device=device_index, collect_stats=False)
                r1.set_scene(rrt_amd.Scene(SCENE, W, H))
                whole = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                r1.render_device(whole.data_ptr(), stream)
                torch.cuda.synchronize()
                n1 = 2
                t0 = time.perf_counter()
                for _ in range(n1):
                    r1.render_device(whole.data_ptr(), stream)
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t0) / n1 * 1e3
                one_gpu = {"ms_per_step": round(ms1, 3), "value": round(W * H * spp / (ms1 * 1e-3) / 1e6, 2), "unit": "Msamples/s", "steps": n1,
                           "note": "the whole frame of this line's workload on rank 0's GPU alone, timed inside the same job while the other ranks wait"}
                r1.close()
                del whole
            except Exception as e:
                one_gpu = {"error": repr(e)}
        barrier()

    # ---- N > 1 on configuration 5: the same job ALSO splits configuration 3 (north_star's "1200x800 spp=500 ... >= 7.5 x at 8
    # GPUs" is stated on C3), so that one driver run reports both: shards of C3 + gather, and C3 whole on rank 0's GPU alone
    strong_c3 = None
    if world > 1 and wl_name == "C5" and not args.no_strong_c3:
        w3, h3, s3 = C3
        if args.spp > 0:
            s3 = args.spp  # (non-headline runs: rehearsals, tests)
        s3r = ShardedRenderer(SCENE, w3, h3, s3, DEPTH, fp64=False, tile_rows=args.tile_rows, device=dev, collect_stats=False)
        n3 = max(args.steps, 10)
        e3, st3 = timed(s3r, n3, 2)
        k3, _ = timed(s3r, n3, 0, gather=False)
        strong_c3 = {"workload": "C3: scenes/final.txt %dx%d spp=%d d=%d fp32, list scan, row-tile shards x%d + one gather per step" % (w3, h3, s3, DEPTH, world), "steps": n3,
                     "ms_per_step": round(e3 / n3 * 1e3, 3), "kernel_only_ms_per_step": round(k3 / n3 * 1e3, 3), "value": round(w3 * h3 * s3 / (e3 / n3) / 1e6, 2), "unit": "Msamples/s"}
        del s3r
        barrier()
        if rank == 0:
            try:
                r1 = rrt_amd.Rrt(w3, h3, s3, DEPTH, use_bvh=False, fp64=False, device=device_index, collect_stats=False)
                r1.set_scene(rrt_amd.Scene(SCENE, w3, h3))
                whole = torch.empty((h3, w3, 3), dtype=torch.float32, device=dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                r1.render_device(whole.data_ptr(), stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n3):
                    r1.render_device(whole.data_ptr(), stream)
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t0) / n3 * 1e3
                strong_c3["one_gpu_ms_per_step"] = round(ms1, 3)
                strong_c3["speedup_vs_one_gpu"] = round(ms1 / (e3 / n3 * 1e3), 3)
                strong_c3["north_star_target"] = ">= 7.5 at 8 GPUs"
                r1.close()
                del whole
            except Exception as e:
                strong_c3["one_gpu_error"] = repr(e)
        barrier()

    # per-rank kernel statistics -> whole-job numbers
    vec = torch.tensor([float(st["bytes_algorithmic"]), st["kernel_ms_sum"] / max(1, st["renders"]), float(st["segments"]), float(st["prim_tests"]), float(st["scanned_segments"]),
                        float(st["candidates"])], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        allv = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(allv, vec)
    else:
        allv = [vec]
    if rank == 0:
        total_bytes = sum(float(v[0]) for v in allv)
        kernel_ms = max(float(v[1]) for v in allv)  # slowest rank's mean launch duration
        segments = sum(float(v[2]) for v in allv)
        prim_tests = sum(float(v[3]) for v in allv)
        scanned = sum(float(v[4]) for v in allv)
        candidates = sum(float(v[5]) for v in allv)
        samples = W * H * spp
        ms_per_step = elapsed / args.steps * 1e3
        logical = total_bytes / (kernel_ms * 1e-3) / 1e9 / world  # GB/s per GPU

        # ---- the binding unit: the vector issue port, which the matrix instructions of the scan's filter share with everything else
        clock = CLOCK_HZ
        roof = {"bound": "valu_issue", "unit": "VALU wave-instructions/clk/SIMD (a matrix instruction = %d)" % MFMA_ISSUE, "peak": VALU_PEAK, "kernel": "rrtx::" + LIST_KERNEL + ", ...>", "kernel_ms": round(kernel_ms, 3),
                "note": "the SIMD's one vector issue port is what the kernel's two kinds of work share: the scan filter (two chained v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays, the results' signs "
                        "shifted into a mask: 1.2 vector instructions per sphere and wave where rounds 1 - 2 spent 8) and the per-lane rest of a path tracer (camera rays, exact tests, shading); rounds 1 - 2 "
                        "reported 0.78 of this peak at 6 200 Msamples/s - fewer instructions, not a busier port, is where the time went; `mfma` prices the matrix instructions alone"}
        # what the run can count itself: the filter's matrix instructions - per scanned wave-segment (>= scanned segments / 64) and block of 32
        # spheres, 2 tiles of 32 rays x 2 halves of the 32 terms
        n_blocks = (488 + MF_BLOCK - 1) // MF_BLOCK
        mfma_min = scanned / world / 64.0 * n_blocks * 4
        if pmc and pmc.get("list_scan"):
            per_clk, lanes, ghz = valu_numbers(pmc["list_scan"])
            if ghz:
                clock = ghz * 1e9
            if pmc["list_scan"].get("kernel_name"):
                roof["kernel"] = pmc["list_scan"]["kernel_name"].split("(")[0].replace("void ", "")
            cyc = pmc["list_scan"]["GRBM_GUI_ACTIVE"] / 8.0
            mm = pmc.get("list_scan_mfma") or {}
            n_mfma = mm.get("SQ_INSTS_MFMA")
            raw_per_clk = per_clk  # SQ_INSTS_VALU / clk / SIMD as counted (a matrix instruction counts once)
            mfma_per_clk = None
            if n_mfma is not None:
                # SQ_INSTS_VALU counts a matrix instruction once, but it holds the port for MFMA_ISSUE plain instructions' clocks: a MODEL, from two
                # separate passes - each rate is formed with ITS OWN pass's clock count (runs differ in clock and duration), then added
                mfma_per_clk = n_mfma / (N_SIMD * (mm["GRBM_GUI_ACTIVE"] / 8.0))
                per_clk = raw_per_clk + (MFMA_ISSUE - 1) * mfma_per_clk
            roof.update({"achieved": round(per_clk, 4), "frac": round(min(1.0, per_clk / VALU_PEAK), 4), "modelled": n_mfma is not None,
                         "measured": {"valu_instructions_per_clk_per_simd": round(raw_per_clk, 4), "mfma_instructions_per_clk_per_simd": round(mfma_per_clk, 5) if mfma_per_clk is not None else None,
                                      "mfma_issue_weight": MFMA_ISSUE,
                                      "note": "achieved = valu + (weight - 1) x mfma: the two rates are MEASURED (separate PMC passes, each over its own GRBM_GUI_ACTIVE / 8), the weight - a matrix "
                                              "instruction holds the vector issue port for 8 clocks, a plain one for 2 (MI355X_MICROARCH.md, cycle constants) - is a constant of the model; frac is clamped to 1"},
                         "valu_lane_utilisation": round(lanes, 4) if lanes else None,
                         "shader_clock_GHz": round(ghz, 3) if ghz else None, "counters": {k: v for k, v in pmc["list_scan"].items() if k != "kernel_name"}, "source": pmc["source"], "kernel_source_sha": pmc.get("kernel_source_sha")})
            if n_mfma is not None:
                mcyc = mm["GRBM_GUI_ACTIVE"] / 8.0
                tf = n_mfma * MFMA_FLOP / (mm.get("duration_ms", kernel_ms) * 1e-3) / 1e12
                roof["mfma"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS, "achieved": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "instructions": int(n_mfma),
                                "pipe_busy_frac": round(mm["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * mcyc), 4) if mm.get("SQ_VALU_MFMA_BUSY_CYCLES") else None, "counters": mm,
                                "note": "v_mfma_f32_32x32x16_f16 of the scan filter (phase 1: a third of the kernel's cycles; the rest of a path tracer's loop - camera rays, exact tests, shading - has no matrix form); "
                                        "instructions x %d flop / kernel time against the dense f16 peak" % MFMA_FLOP}
            roof["traffic"] = (pmc["hbm_read_bytes"] + pmc["hbm_write_bytes"]) if "hbm_read_bytes" in pmc else None
            if "hbm_read_bytes" in pmc:
                roof["traffic_detail"] = {"read_bytes": pmc["hbm_read_bytes"], "write_bytes": pmc["hbm_write_bytes"], "hbm_GBs": round((pmc["hbm_read_bytes"] + pmc["hbm_write_bytes"]) / (kernel_ms * 1e-3) / 1e9, 1),
                                          "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, separate passes; per launch"}
        else:
            roof.update({"achieved": None, "frac": None, "traffic": None, "source": "no PMC pass possible in this run and no committed profile of this device code"})
        filt = mfma_min * MFMA_ISSUE / (kernel_ms * 1e-3 * N_SIMD * clock)
        roof["filter_only"] = {"achieved": round(filt, 4), "frac": round(filt / VALU_PEAK, 4), "mfma_instructions_min": int(mfma_min), "mfma_TFLOPs_min": round(mfma_min * MFMA_FLOP / (kernel_ms * 1e-3) / 1e12, 1),
                               "note": "lower bound counted by the kernel itself: scanned segments / 64 lanes x %d blocks of 32 spheres x 4 matrix instructions (x %d issue slots each), over kernel time x 1024 SIMDs x clock" % (n_blocks, MFMA_ISSUE)}
        if roof.get("frac") is None:
            roof["achieved"], roof["frac"] = roof["filter_only"]["achieved"], roof["filter_only"]["frac"]
            roof["source"] += "; achieved / frac = the filter-only lower bound"
        roof["logical_hbm"] = {"achieved": round(logical, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(logical / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(total_bytes / world),
                               "note": "SURVEY.md 8(d): what the REFERENCE's list scan reads (segments x 488 spheres x 16 B + frame) / kernel time against HBM3E's 8 TB/s. Not a physical fraction: "
                                       "the 7.8 KB scene is read from the scalar cache / LDS, one record read serves the 64 rays of a wave, camera rays are resolved from per-pixel lists"}
        line = {
            "metric": "Msamples/s (WxHxspp) on scenes/final.txt fp32",
            "value": round(samples / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: scenes/final.txt %dx%d spp=%d d=%d fp32, list scan of 488 spheres (-b; camera rays via per-pixel candidate lists, every other segment through the whole list)" % (wl_name, W, H, spp, DEPTH),
                       "parallelism": "row-tile shards x%d (tile_rows=%d)%s" % (world, args.tile_rows, ", one RCCL gather to rank 0 per step" if world > 1 else ""),
                       "sample_chunk": st["sample_chunk"], "segments_per_sample": round(segments / samples, 4), "prim_tests_per_launch": int(prim_tests),
                       "prim_tests_executed_per_launch": int(scanned * 488 + candidates), "scanned_segments_per_sample": round(scanned / samples, 4),
                       "sky_pixels": int(st.get("sky_pixels", 0)), "first_bounce_prepass": int(st.get("first_bounce", 0))},
            "roofline": roof,
        }
        if kernel_only is not None:
            line["kernel_only_ms_per_step"] = round(kernel_only, 3)
            line["gather_ms_per_step"] = round(max(0.0, ms_per_step - kernel_only), 3)
        if rccl_info is not None:
            line["config"]["rccl"] = rccl_info
        if one_gpu is not None:
            line["one_gpu_same_workload"] = one_gpu
        if strong_c3 is not None:
            line["strong_c3"] = strong_c3
        if accel is not None:
            line["accelerated"] = accel
        if configs is not None:
            line["configs"] = configs
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # never lose the GPU measurement to a CPU-side hiccup
                line["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
