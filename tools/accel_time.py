"""Developer script (GPU box): kernel time of the accelerated mode (and the list scan) on final.txt, plus an image check
against the list scan.  args: [spp ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
spps = [int(x) for x in sys.argv[1:]] or [8, 48, 500]
for fp64 in (False, True):
    s = rrt_amd.Scene(scene_path("final"), W, H, fp64=fp64)
    for spp in spps:
        out = {}
        for bvh in (False, True):
            r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, fp64=fp64)
            fb = r.render(s)
            t = []
            for _ in range(3):
                r.render()
                t.append(r.stats["kernel_ms"])
            out[bvh] = (fb, min(t), r.stats["segments"])
            r.close()
        same = np.array_equal(out[False][0], out[True][0]) and out[False][2] == out[True][2]
        print("%s spp %4d  list %.3f ms  accel %.3f ms  (%.1f Msamples/s)  identical=%s" % ("f64" if fp64 else "f32", spp, out[False][1], out[True][1], W * H * spp / out[True][1] / 1e3, same), flush=True)
