"""Developer script (GPU box): SURVEY.md 8(f) N2 - a triangle mesh through the list scan and through the grid.
  python tools/mesh_check.py [nu nv] [W H spp]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from rrt_amd import _lib
from _oracle import mesh_scene
a = [int(x) for x in sys.argv[1:]]
nu, nv = (a[0], a[1]) if len(a) >= 2 else (16, 32)
W, H, spp = (a[2], a[3], a[4]) if len(a) >= 5 else (600, 400, 16)
f, n_tri = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), nu, nv)
for fp64 in (True, False):
    sc = rrt_amd.Scene(f, W, H, fp64=fp64)
    out = {}
    for bvh in (False, True):
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, fp64=fp64); r.render(sc); out[bvh] = r.render(sc); st = r.stats; r.close()
        print("%s %d triangles %dx%d spp %d  use_bvh=%d: %.3f ms, grid cells %d, scanned segments %d of %d" % ("fp64" if fp64 else "fp32", n_tri, W, H, spp, bvh, st["kernel_ms"], st["accel_cells"], st["scanned_segments"], st["segments"]), flush=True)
    print("   identical:", np.array_equal(out[False], out[True]), flush=True)
    if fp64:
        r = rrt_amd.Rrt(W, H, min(spp, 4), 50, use_bvh=True, fp64=True, flags=_lib.FLAG_VERIFY_LISTS); r.render(sc); st = r.stats; r.close()
        print("   walk mismatches %d of %d segments" % (st["list_mismatches"], st["segments"]), flush=True)
