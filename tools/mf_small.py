"""Developer script (GPU box): where the filter on the matrix cores starts to pay - scenes of n spheres, list scan, both filters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, 10
def timed(f, flags):
    s = rrt_amd.Scene(f, W, H)
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, flags=flags)
    fb = r.render(s)
    t = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(3))
    m = r.stats["scan_mfma"]
    r.close()
    return fb, t, m
a = timed(scene_path("test1"), rrt_amd.FLAG_SCAN_NO_MFMA); b = timed(scene_path("test1"), 0)
print("test1.txt (C2): valu %.3f ms, mfma(%d) %.3f ms, identical=%s" % (a[1], b[2], b[1], np.array_equal(a[0], b[0])), flush=True)
rng = np.random.default_rng(1)
for n in (4, 12, 16, 28, 32, 48, 64, 96, 128, 256):
    k = int(np.ceil(np.sqrt(n)))
    lines = ["camera 13 2 3 0 0 0 0 1 0 20 0.1 10", "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.1", "material g dielectric 1.5", "sphere 0 -1000 0 1000 a"]
    lines += ["sphere %r 0.2 %r 0.2 %s" % (float(i % k - k / 2 + 0.6 * rng.random()), float(i // k - k / 2 + 0.6 * rng.random()), "amg"[i % 3]) for i in range(n - 1)]
    f = "/tmp/mf_small_%d.txt" % n
    open(f, "w").write("\n".join(lines) + "\n")
    a = timed(f, rrt_amd.FLAG_SCAN_NO_MFMA); b = timed(f, 0)
    print("%4d spheres: valu %.3f ms, mfma(%d) %.3f ms, identical=%s" % (n, a[1], b[2], b[1], np.array_equal(a[0], b[0])), flush=True)
