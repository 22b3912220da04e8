"""Developer script (GPU box): kernel time vs spp on final.txt -> fixed overhead and slope."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
s = rrt_amd.Scene(scene_path("final"), W, H)
xs, ys = [], []
for spp in (8, 16, 48, 104, 200, 504, 1000):
    r = rrt_amd.Rrt(W, H, spp, 50)
    r.render(s)
    t = []
    for _ in range(3):
        r.render()
        t.append(r.stats["kernel_ms"])
    print("spp %4d  chunk %d  kernel %.3f ms (min of 3: %s)  %.1f Msamples/s" % (spp, r.stats["sample_chunk"], min(t), ["%.2f" % v for v in t], W * H * spp / min(t) / 1e3), flush=True)
    xs.append(spp); ys.append(min(t))
    r.close()
A = np.polyfit(xs[2:], ys[2:], 1)
print("fit (spp >= 48): %.4f ms/spp + %.3f ms fixed" % (A[0], A[1]))
