"""Device-resident rendering through torch (the path bench.py and the multi-GPU gather use): the
framebuffer stays in HBM, the kernel runs on torch's current stream."""
import numpy as np
import pytest

from _oracle import Oracle, scene_path

pytestmark = pytest.mark.gpu


def _list_rrt(gpu, *args, **kw):
    """Rrt on the list scan (`-b`): what most tests here exercise.  The accelerated closest hit (use_bvh,
    the CLI's default) has its own tests; its images must equal these bit for bit."""
    kw.setdefault("use_bvh", False)
    return gpu.Rrt(*args, **kw)


def test_render_into_a_torch_tensor_on_the_current_stream(gpu):
    import torch

    from rrt_amd.dist import ShardedRenderer

    w, h, spp = 64, 40, 4
    sr = ShardedRenderer(scene_path("final"), w, h, spp, sample_chunk=-1)
    assert sr.world == 1 and len(sr.rows) == h
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        frame = sr.render()
        side.synchronize()
    fo, _ = Oracle(scene_path("final"), w, h, False).render(spp, 50, 1984, order=1)
    assert np.array_equal(frame.cpu().numpy(), fo)
    st = sr.rrt.collect()
    assert st["renders"] == 1 and st["kernel_ms"] > 0


def test_back_to_back_launches_are_independent(gpu):
    import torch

    from rrt_amd.dist import ShardedRenderer

    sr = ShardedRenderer(scene_path("test2"), 80, 50, 5)
    first = sr.render_local().clone()
    for _ in range(3):
        sr.render_local()
    torch.cuda.synchronize()
    assert torch.equal(first, sr.local)
    st = sr.rrt.collect()
    assert st["renders"] == 4 and st["kernel_ms_sum"] >= st["kernel_ms"]


def _rank_main(rank, world, port, out_dir, w, h, spp, tile_rows):
    import os
    import sys

    import torch
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _oracle import scene_path as sp
        from rrt_amd.dist import ShardedRenderer

        torch.cuda.set_device(0)  # the test box has one GPU: the ranks share it, the gather is staged
        sr = ShardedRenderer(sp("final"), w, h, spp, tile_rows=tile_rows, device="cuda:0")
        assert sr.world == world and sr.rank == rank
        frame = sr.render(dst=0)
        torch.cuda.synchronize()
        if rank == 0:
            import numpy as np

            np.save(os.path.join(out_dir, "frame.npy"), frame.cpu().numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows", [(2, 4), (3, 5)])
def test_sharded_render_across_ranks_equals_the_full_frame(gpu, tmp_path, world, tile_rows):
    """N > 1 with the real kernels: `world` processes (sharing this box's one GPU) render their row
    tiles and gather to rank 0; the assembled frame must be the single-process frame bit for bit."""
    import socket

    import torch.multiprocessing as mp

    w, h, spp = 72, 50, 12
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), w, h, spp, tile_rows), nprocs=world, join=True)
    got = np.load(tmp_path / "frame.npy")
    r = _list_rrt(gpu, w, h, spp, 50)
    want = r.render(gpu.Scene(scene_path("final"), w, h))
    r.close()
    assert np.array_equal(got, want)
    assert np.array_equal(want, Oracle(scene_path("final"), w, h, False).render(spp, 50, 1984, order=1, chunk=8)[0])


def test_multi_gpu_front_end_writes_the_same_png(gpu, tmp_path):
    """python -m rrt_amd.dist: the reference's flags, one process per GPU, row-tile shards + one gather.
    Launched with 3 ranks (sharing this box's GPU, gather staged through gloo) and with none; both
    PNGs must equal the single-context image, in both closest-hit modes."""
    import os
    import subprocess
    import sys

    from PIL import Image

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w, h, spp = 120, 90, 6
    sc = gpu.Scene(scene_path("final"), w, h)
    r = _list_rrt(gpu, w, h, spp, 50)
    want = gpu.quantise(r.render(sc), spp)
    r.close()
    env = dict(os.environ, RRTX_DIST_BACKEND="gloo", PYTHONPATH=root)
    for i, extra in enumerate(([], ["-b"])):
        out = str(tmp_path / ("multi%d.png" % i))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(29631 + i), "-m", "rrt_amd.dist", "-i",
               scene_path("final"), "-o", out, "-w", str(w), "-h", str(h), "-s", str(spp), "-T", "8"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert p.stderr.count("took ") == 3
        assert np.array_equal(np.asarray(Image.open(out)), want), extra
    out = str(tmp_path / "single.png")
    p = subprocess.run([sys.executable, "-m", "rrt_amd.dist", "-i", scene_path("final"), "-o", out, "-w", str(w), "-h", str(h), "-s", str(spp)], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert np.array_equal(np.asarray(Image.open(out)), want)


def test_bench_line_of_a_two_rank_run(gpu):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed with two ranks over
    gloo on this box's GPU (RRTX_BENCH_BACKEND=gloo: the ranks share the device, the gather is staged through the host - the
    plumbing is what is checked, the numbers mean nothing): ONE JSON line on stdout with the contract's fields, the 8-GPU
    workload of BASELINE.json (configuration 5) as the default for N > 1, kernel-only time beside the step time."""
    import json
    import os
    import subprocess
    import sys

    from _oracle import ROOT

    env = dict(os.environ, RRTX_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]  # one line, nothing else on stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["warmup"] == 0 and j["unit"] == "Msamples/s" and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert j["scaling"] == "strong" and "3840x2160" in j["config"]["workload"] and "x2" in j["config"]["parallelism"]
    assert abs(j["value"] - 3840 * 2160 * 24 / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * j["value"]
    assert j["kernel_only_ms_per_step"] > 0 and j["gather_ms_per_step"] >= 0
    assert j["roofline"]["bound"] == "valu_issue" and 0 < j["roofline"]["frac"] <= 1 and "logical_hbm" in j["roofline"]
    assert "cpu_baseline" not in j  # rank 0 at N = 1 only
    rc = j["config"]["rccl"]  # what the process group saw: the driver is the only one who can run N > 1 on real devices
    assert rc["ranks"] == 2 and rc["backend"] == "gloo" and len(rc["devices"]) == 2 and [d["rank"] for d in rc["devices"]] == [0, 1]
    assert all(d["name"] and d["ordinal"] == 0 for d in rc["devices"]) and rc["distinct_devices"] == 1  # (a rehearsal: both ranks on this box's one GPU)
    c3 = j["strong_c3"]  # north_star's 7.5 x is stated on configuration 3: split the same way by the same job, next to C3 whole on rank 0's GPU
    assert "1200x800" in c3["workload"] and c3["ms_per_step"] >= c3["kernel_only_ms_per_step"] > 0 and c3["one_gpu_ms_per_step"] > 0 and c3["speedup_vs_one_gpu"] > 0
    assert abs(c3["value"] - 1200 * 800 * 24 / (c3["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * c3["value"]
    one = j["one_gpu_same_workload"]  # the same frame on rank 0's GPU alone, inside the same job
    assert "error" not in one and one["ms_per_step"] > 0 and abs(one["value"] - 3840 * 2160 * 24 / (one["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * one["value"]
