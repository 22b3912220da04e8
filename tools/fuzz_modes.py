"""Developer script (GPU box): random launch shapes and tuning parameters, list scan against the grid mode
(hand-off / resume pass included) on the shipped scenes: every image pair must be identical.
  python tools/fuzz_modes.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
import tempfile
from _oracle import mesh_scene
_tmp = tempfile.mkdtemp()
_mesh = mesh_scene(os.path.join(_tmp, "mesh.txt"), 12, 24)[0]
_rng0 = np.random.default_rng(9)
_many = os.path.join(_tmp, "many.txt")
open(_many, "w").write("\n".join(["camera 10 3 10 0 0 0 0 1 0 35 0.05 14", "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.05", "material g dielectric 1.5", "sphere 0 -1000 0 1000 a"] +
                                 ["sphere %r %r %r %r %s" % (float(x), float(y), float(z), float(r), "amg"[k % 3]) for k, (x, y, z, r) in enumerate(zip(_rng0.uniform(-30, 30, 3999), _rng0.uniform(0.1, 0.3, 3999), _rng0.uniform(-30, 30, 3999), _rng0.uniform(0.05, 0.3, 3999)))]) + "\n")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n):
    name = ["final", "final", "final", "test2", "test3", "xform", "mesh", "many"][int(rng.integers(8))]
    fp64 = bool(rng.integers(2)) or name == "mesh"  # (the mesh is gridded in fp64 only)
    w, h = int(rng.integers(8, 400)), int(rng.integers(8, 260))
    if name in ("mesh", "many"):
        w, h = w // 3 + 8, h // 3 + 8  # (their list scans are slow)
    spp = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 31, 50]))
    kw = dict(sample_chunk=int(rng.choice([0, -1, 1, 3, 8, 16])), handoff_lanes=int(rng.choice([0, 1, 7, 40, 64])), handoff_iters=int(rng.choice([0, 1, 3, 8, 50])))
    shards = int(rng.choice([1, 1, 2, 3]))
    kw["tile_rows"] = int(rng.choice([1, 4, 8]))
    depth = int(rng.choice([50, 50, 5, 1]))
    path = {"xform": os.path.join(ROOT, "tests", "golden", "scenes", "xform.txt"), "mesh": _mesh, "many": _many}.get(name) or scene_path(name)
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
