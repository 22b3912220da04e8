"""The N > 1 path on CPU: two processes over gloo shard a frame by interleaved row tiles, each
produces its rows (the oracle stands in for the device renderer here - this tier has no GPU) and
rank 0 gathers.  The assembled frame must equal the unsharded one bit for bit, for any tiling."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rrt_amd.dist import gather_frame, shard_rows

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tile_rows, height, width, out_dir):
    sys.path.insert(0, HERE)
    from _oracle import Oracle, scene_path

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = Oracle(scene_path("test2"), width, height, False)
        rows = shard_rows(height, rank, world, tile_rows)
        block = np.zeros((len(rows), width, 3), dtype=np.float32)
        for k, j in enumerate(rows):  # each rank renders ONLY its own rows
            fb, _ = o.render(2, 50, 1984, order=1, rows=(int(j), int(j) + 1))
            block[k] = fb[j]
        frame = gather_frame(torch.from_numpy(block), height, tile_rows, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, "frame.npy"), frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows", [1, 4, 5])
def test_two_rank_gather_equals_unsharded_frame(tmp_path, tile_rows):
    from _oracle import Oracle, scene_path

    height, width, world = 22, 30, 2  # 22 rows: ragged last tile, unequal row counts per rank
    mp.spawn(_worker, args=(world, _free_port(), tile_rows, height, width, str(tmp_path)), nprocs=world, join=True)
    full, _ = Oracle(scene_path("test2"), width, height, False).render(2, 50, 1984, order=1)
    assert np.array_equal(np.load(tmp_path / "frame.npy"), full)


def test_shard_rows_partition_the_frame():
    for h in (1, 7, 22, 800, 2160):
        for world in (1, 2, 3, 4, 8):
            for t in (1, 2, 4, 7, 16):
                parts = [shard_rows(h, r, world, t) for r in range(world)]
                allrows = np.sort(np.concatenate(parts))
                assert np.array_equal(allrows, np.arange(h))
                for p in parts:
                    assert np.all(np.diff(p) > 0) if len(p) > 1 else True
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= t


def test_native_gather_plan_matches_the_definition_and_dist_shard_rows(tmp_path):
    """rrtx_group's side of the multi-GPU path that no one-GPU box can run: the row-tile plan the library
    compiles - how many rows a member renders, which frame row each of its local rows is, where a frame row lies in the
    gathered buffer (rrt_amd/csrc/rrtx_device.h, shared by rrtx_api.cpp, the render kernel's comments and
    deinterleave_kernel) - built for the host and checked for N = 1 .. 9 members, tiles of 1 .. 16 rows, ragged last tiles
    and more members than tiles (1008 cases): shards disjoint and covering, counts equal to the definition, gather ->
    de-interleave = identity; and every case's row lists equal rrt_amd.dist.shard_rows, the rule of the
    one-process-per-GPU path."""
    import subprocess

    import numpy as np

    from rrt_amd.dist import shard_rows
    from _oracle import ROOT

    exe = str(tmp_path / "gather_plan_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "gather_plan_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "print"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "1008 cases ok" in r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1008
    for line in lines:
        head, body = line.split(":")
        H, T, N = (int(x) for x in head.split())
        parts = body.split("|")
        assert len(parts) == N
        for rank, part in enumerate(parts):
            assert np.array_equal(np.array(part.split(), dtype=np.int64), shard_rows(H, rank, N, T)), (H, T, N, rank)
