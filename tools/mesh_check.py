"""Developer script (GPU box): SURVEY.md 8(f) N2 - a triangle mesh through the list scan and through the grid.
  python tools/mesh_check.py [nu nv] [W H spp]
fp64: the grid is proven (images must be identical, 0 walk mismatches).  fp32: triangles are gridded under the
approximate rule (rrtx_grid.h): prints the fraction of segments the walk resolves differently from the sequential
scan (VERIFY build) and how many pixels of the frame differ."""
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
        print("%s %d triangles %dx%d spp %d  use_bvh=%d: %.3f ms, grid cells %d (exact %d), scanned segments %d of %d" % ("fp64" if fp64 else "fp32", n_tri, W, H, spp, bvh, st["kernel_ms"], st["accel_cells"], st["accel_exact"], st["scanned_segments"], st["segments"]), flush=True)
    diff = (out[False] != out[True]).any(axis=2)
    q0, q1 = rrt_amd.quantise(out[False], spp).astype(int), rrt_amd.quantise(out[True], spp).astype(int)
    print("   identical: %s (%d of %d pixels differ; 8-bit image: %d channels differ, max %d LSB)" % (not diff.any(), int(diff.sum()), W * H, int((q0 != q1).sum()), int(np.abs(q0 - q1).max())), flush=True)
    vs = min(spp, 2)
    r = rrt_amd.Rrt(W, H, vs, 50, use_bvh=True, fp64=fp64, flags=_lib.FLAG_VERIFY_LISTS); r.render(sc); st = r.stats; r.close()
    print("   walk mismatches %d of %d segments (%.2e)" % (st["list_mismatches"], st["segments"], st["list_mismatches"] / max(1, st["segments"])), flush=True)
