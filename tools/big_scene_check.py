"""Developer script (GPU box): scenes of N small spheres (tables beyond the LDS copies), list scan and grid.
  python tools/big_scene_check.py N [W H spp]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
a = [int(x) for x in sys.argv[1:]]
N = a[0] if a else 4000
W, H, spp = (a[1], a[2], a[3]) if len(a) >= 4 else (1200, 800, 16)
rng = np.random.default_rng(9)
half = float(0.5 * np.sqrt(N))
lines = ["camera %r 3 %r 0 0 0 0 1 0 35 0.05 %r" % (0.33 * half, 0.33 * half, 0.45 * half), "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.05", "material g dielectric 1.5", "sphere 0 -1000 0 1000 a"]
for k in range(N - 1):
    x, z = rng.uniform(-half, half, 2)
    lines.append("sphere %r %r %r %r %s" % (float(x), float(rng.uniform(0.1, 0.3)), float(z), float(rng.uniform(0.05, 0.3)), "amg"[k % 3]))
f = os.path.join(tempfile.mkdtemp(), "many.txt")
open(f, "w").write("\n".join(lines) + "\n")
sc = rrt_amd.Scene(f, W, H)
out = {}
for bvh in (False, True):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh); r.render(sc); out[bvh] = r.render(sc); st = r.stats; r.close()
    print("%d spheres %dx%d spp %d use_bvh=%d: %.3f ms = %.1f Msamples/s, %.1f G tests/s (algorithmic), grid cells %d, scanned %d of %d segments" % (
        N, W, H, spp, bvh, st["kernel_ms"], W * H * spp / st["kernel_ms"] / 1e3, st["segments"] * N / st["kernel_ms"] / 1e6, st["accel_cells"], st["scanned_segments"], st["segments"]), flush=True)
print("identical:", np.array_equal(out[False], out[True]))
