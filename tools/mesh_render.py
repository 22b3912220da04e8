"""Developer script (GPU box; for rocprofv3 --kernel-trace): the 27 072-triangle mesh scene rendered a few times.
args: fp64 use_bvh [nu nv W H spp]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import mesh_scene
fp64, bvh = int(sys.argv[1]), int(sys.argv[2])
nu, nv, W, H, spp = [int(x) for x in sys.argv[3:8]] if len(sys.argv) >= 8 else (48, 96, 600, 400, 16)
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), nu, nv)
sc = rrt_amd.Scene(f, W, H, fp64=bool(fp64))
r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bool(bvh), fp64=bool(fp64))
for k in range(4):
    r.render(sc if k == 0 else None)
print(n, "triangles", r.stats["kernel_ms"], "ms", r.stats["accel_cells"], "cells", r.stats["segments"], "segments")
