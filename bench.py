#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the render hot path on scenes/final.txt (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over the workload: render the 1200x800 depth=50 fp32 frame of
scenes/final.txt (scene tables already resident in HBM, framebuffer left in HBM) and, for N > 1, the
one gather of the row-tile shards to rank 0 over RCCL.  N = 1 is BASELINE.json's configuration, spp = 500.
N > 1 keeps the work per GPU fixed ("scaling": "weak", as the frame is a partitioned path): the same
frame, row tiles dealt round-robin to the ranks, at spp = 500 x N — every rank traces 480 M samples, as
BASELINE.json's own 8-GPU configuration scales the job (3840x2160 spp 1000) rather than splitting the
1-GPU one.  `--strong` shards the spp = 500 frame instead (9.7 ms of work per GPU at N = 8; DESIGN.md has
the fixed per-launch cost that then shows).  The image is bit-identical for every N and tile size.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline     - logical primitive-read roofline of the render kernel: algorithmic bytes per launch
                 (primitive tests x 16 B + framebuffer, counted on the device) / the kernel's mean
                 launch duration (HIP events on the launch stream) against HBM3E's 8 TB/s.  The
                 scene is 7.8 KB and lives in the scalar cache, so frac > 1 is expected; real HBM
                 traffic (PMC) is reported beside it when profiles/ holds a measurement.
  cpu_baseline - the reference's own OpenMP binary (oracle/_ref/rrto, built from the reference
                 sources) timed here on the host cores, on a bounded sample of the same scene.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE = os.path.join(ROOT, "scenes", "final.txt")
WIDTH, HEIGHT, SPP, DEPTH = 1200, 800, 500, 50
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline():
    """Reference CPU build on a bounded sample (~10-30 s of CPU work).  Test infrastructure only."""
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
        out["brute_force_sample"] = "rrto -b (list scan, the mode the HIP kernel implements) %dx%d spp=%d: %.2f s" % (w2, h2, s2, sec_b)
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


def measured_traffic():
    """HBM bytes per launch from a committed rocprofv3 PMC run of this same command (or None)."""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def measured_valu_issue():
    """VALU issue utilisation of the headline kernel from the committed PMC summary (or None)."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_render_kernel.csv")
    try:
        for line in open(p):
            if line.startswith("valu_issue_utilisation,"):
                return float(line.split(",")[1])  # the first block of the file is the list-scan kernel
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (non-headline runs only); default 500 x N (weak scaling)")
    ap.add_argument("--strong", action="store_true", help="N > 1: shard the spp = 500 frame (fixed total work) instead of scaling spp with N")
    ap.add_argument("--tile-rows", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accel", action="store_true", help="skip the extra use_bvh measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible - the render path has no CPU fallback")
    # one rank per GPU.  RRTX_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks (ranks then share devices and the gather is staged through the host): it checks the
    # plumbing, its numbers mean nothing.
    backend = os.environ.get("RRTX_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from rrt_amd.dist import ShardedRenderer

    if args.spp <= 0:
        args.spp = SPP if args.strong else SPP * world

    sr = ShardedRenderer(SCENE, WIDTH, HEIGHT, args.spp, DEPTH, fp64=False, tile_rows=args.tile_rows, device=torch.device("cuda", device_index), collect_stats=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sr.render()
    barrier()
    sr.rrt.collect()  # drop warm-up launches from the event statistics
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sr.render()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = sr.rrt.collect()

    # the same steps with use_bvh (the CLI's default): reported beside the headline, which
    # stays the list scan the north star names and the roofline is defined for
    accel = None
    if not args.no_accel:
        sa = ShardedRenderer(SCENE, WIDTH, HEIGHT, args.spp, DEPTH, fp64=False, tile_rows=args.tile_rows, device=torch.device("cuda", device_index), collect_stats=True, use_bvh=True)
        sa.render()
        barrier()
        sa.rrt.collect()
        barrier()
        ta = time.perf_counter()
        for _ in range(args.steps):
            sa.render()
        barrier()
        elapsed_a = time.perf_counter() - ta
        if world > 1:
            tt = torch.tensor([elapsed_a], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed_a = float(tt.item())
        sta = sa.rrt.collect()
        accel = {"value": round(WIDTH * HEIGHT * args.spp / (elapsed_a / args.steps) / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(elapsed_a / args.steps * 1e3, 3),
                 "kernel_ms": round(sta["kernel_ms_sum"] / max(1, sta["renders"]), 3), "grid_cells": sta["accel_cells"],
                 "note": "use_bvh = 1: closest hit through a uniform grid + always-list, exact test and tie rules of the list scan, image bit-identical (tests/test_gpu_parity.py); "
                         "its segments fall back to the list scan only for rays outside the grid's proven range (DESIGN.md)"}
        del sa

    # per-rank kernel statistics -> whole-job roofline numbers
    vec = torch.tensor([float(st["bytes_algorithmic"]), st["kernel_ms_sum"] / max(1, st["renders"]), float(st["segments"]), float(st["prim_tests"]), float(st["scanned_segments"]),
                        float(st["candidates"])], dtype=torch.float64,
                       device="cuda" if backend == "nccl" else "cpu")
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
        samples = WIDTH * HEIGHT * args.spp
        ms_per_step = elapsed / args.steps * 1e3
        achieved = total_bytes / (kernel_ms * 1e-3) / 1e9 / world  # GB/s per GPU
        line = {
            "metric": "Msamples/s (WxHxspp) on scenes/final.txt fp32",
            "value": round(samples / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "scenes/final.txt %dx%d spp=%d d=%d fp32, brute-force list scan (488 spheres)" % (WIDTH, HEIGHT, args.spp, DEPTH), "parallelism": "row-tile shards x%d (tile_rows=%d)%s" % (world, args.tile_rows, ", RCCL gather to rank 0" if world > 1 else ""),
                       "sample_chunk": st["sample_chunk"], "segments_per_sample": round(segments / samples, 4), "prim_tests_per_launch": int(prim_tests),
                       "prim_tests_executed_per_launch": int(scanned * 488 + candidates), "scanned_segments_per_sample": round(scanned / samples, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic() if world == 1 and args.spp == SPP else None,
                         "kernel": "rrtx::render_kernel<float, true, 1, false, 0, false>", "kernel_ms": round(kernel_ms, 3), "algorithmic_bytes_per_launch": int(total_bytes / world),
                         "valu_issue": {"filter_frac": round(float(scanned / world * 488) * 8 / 64 / (kernel_ms * 1e-3 * 1024 * 2.4e9 * 0.5), 4), "pmc_frac": measured_valu_issue() if world == 1 and args.spp == SPP else None,
                                        "note": "the unit that binds: wave-instructions per clock per SIMD against the peak of 0.5 (256 CUs x 4 SIMDs at 2.4 GHz); filter_frac counts only the scan filter's 8 instructions per executed (ray, sphere) test of this run, pmc_frac is SQ_INSTS_VALU of the committed rocprofv3 pass (profiles/r01_pmc_render_kernel.csv)"},
                         "note": "logical primitive-read roofline (SURVEY.md 8d): algorithmic bytes = what the reference's list scan reads (segments x 488 spheres x 16 B); a record read from the scalar cache or LDS serves all 64 rays of a wave and camera rays are resolved from per-pixel candidate lists (config.prim_tests_executed_per_launch), so frac > 1 is legitimate; binding unit: VALU issue (DESIGN.md 3)"},
        }
        if accel is not None:
            line["accelerated"] = accel
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # never lose the GPU measurement to a CPU-side hiccup
                line["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
