"""Developer script (GPU box): random launch shapes and tuning parameters, list scan against the grid mode
(hand-off / resume pass included) on the shipped scenes: every image pair must be identical.
  python tools/fuzz_modes.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n):
    name = ["final", "final", "final", "test2", "test3", "xform"][int(rng.integers(6))]
    fp64 = bool(rng.integers(2))
    w, h = int(rng.integers(8, 400)), int(rng.integers(8, 260))
    spp = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 31, 50]))
    kw = dict(sample_chunk=int(rng.choice([0, -1, 1, 3, 8, 16])), handoff_lanes=int(rng.choice([0, 1, 7, 40, 64])), handoff_iters=int(rng.choice([0, 1, 3, 8, 50])))
    shards = int(rng.choice([1, 1, 2, 3]))
    kw["tile_rows"] = int(rng.choice([1, 4, 8]))
    depth = int(rng.choice([50, 50, 5, 1]))
    path = os.path.join(ROOT, "tests", "golden", "scenes", "xform.txt") if name == "xform" else scene_path(name)
    sc = rrt_amd.Scene(path, w, h, fp64=fp64)
    for rank in range(shards):
        imgs = []
        for bvh in (False, True):
            r = rrt_amd.Rrt(w, h, spp, depth, use_bvh=bvh, fp64=fp64, shard_rank=rank, shard_count=shards, **kw)
            imgs.append(r.render(sc).copy())
            st = r.stats
            r.close()
        same = np.array_equal(imgs[0], imgs[1])
        bad += not same
        print("%3d %-6s %s %3dx%-3d spp %-2d d %-2d shard %d/%d %s cells %d -> %s" % (case, name, "f64" if fp64 else "f32", w, h, spp, depth, rank, shards, kw, st["accel_cells"], "same" if same else "DIFFERENT"), flush=True)
print("mismatching pairs:", bad)
sys.exit(1 if bad else 0)
