"""Developer script (GPU box): the worst case for fp32 triangles in the grid - a tessellated PLANE seen at a grazing angle: rays
skim thousands of coplanar triangles, each with |a| = |e1 . (d x e2)| near the reference's 1e-7 cut.  Prints how many segments
the walk resolves differently from the sequential scan (VERIFY build) and how many pixels differ from the list scan's frame."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from rrt_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
W, H, spp = 400, 300, 8
def scene(path, cam):
    rng = np.random.default_rng(4)
    L = [cam, "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.05", "material g dielectric 1.5", "material r lambertian 0.8 0.2 0.2"]
    for k in range(40):
        L.append("sphere %r 0.25 %r 0.25 %s" % (float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)), "amg"[k % 3]))
    xs = np.linspace(-10, 10, n + 1)
    L.append("obj_beg %d %d" % ((n + 1) ** 2, 2 * n * n))
    for i in range(n + 1):
        for j in range(n + 1):
            L.append("obj_vtx %r 0.0 %r" % (float(xs[i]), float(xs[j])))
    for i in range(n):
        for j in range(n):
            a, b, c, d = i * (n + 1) + j, i * (n + 1) + j + 1, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1
            L.append("obj_tri %d %d %d" % (a, b, c)); L.append("obj_tri %d %d %d" % (b, d, c))
    L += ["obj_end", "obj 0 a"]
    open(path, "w").write("\n".join(L) + "\n")
    return path
d = tempfile.mkdtemp()
for name, cam in (("grazing 0.5 deg", "camera -14 0.12 0 0 0.0 0 0 1 0 30 0.0 14"), ("low 5 deg", "camera -12 1.2 0.3 0 0 0 0 1 0 35 0.02 12"), ("ordinary", "camera -9 4 6 0 0 0 0 1 0 40 0.05 11")):
    f = scene(os.path.join(d, "plane.txt"), cam)
    sc = rrt_amd.Scene(f, W, H)
    out = {}
    for bvh in (False, True):
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh); out[bvh] = r.render(sc); st = r.stats; r.close()
        print("%-16s %d triangles use_bvh=%d: %.2f ms, cells %d exact %d" % (name, 2 * n * n, bvh, st["kernel_ms"], st["accel_cells"], st["accel_exact"]), flush=True)
    diff = (out[False] != out[True]).any(axis=2)
    r = rrt_amd.Rrt(W, H, 2, 50, use_bvh=True, flags=_lib.FLAG_VERIFY_LISTS); r.render(sc); st = r.stats; r.close()
    print("   pixels differing %d of %d; walk mismatches %d of %d segments (%.2e)" % (int(diff.sum()), W * H, st["list_mismatches"], st["segments"], st["list_mismatches"] / max(1, st["segments"])), flush=True)
