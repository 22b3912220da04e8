"""Developer script (GPU box): the list scan with the filter on the matrix cores against the same scan with the filter on the
vector unit (RRTX_FLAG_SCAN_NO_MFMA): identical frames and segment counts, kernel times.  args: [spp ...]"""
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
        for flags in (rrt_amd.FLAG_SCAN_NO_MFMA, 0):
            r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, fp64=fp64, flags=flags, collect_stats=True)
            fb = r.render(s)
            st = dict(r.stats)
            r.close()
            r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, fp64=fp64, flags=flags, collect_stats=False)
            r.render(s)
            t = []
            for _ in range(3):
                r.render()
                t.append(r.stats["kernel_ms"])
            out[flags] = (fb, min(t), st["segments"], st["candidates"])
            r.close()
        a, b = out[rrt_amd.FLAG_SCAN_NO_MFMA], out[0]
        same = np.array_equal(a[0], b[0]) and a[2] == b[2]
        print("%s spp %4d  valu %.3f ms (%d candidates)  mfma %.3f ms (%d candidates)  %.1f Msamples/s  identical=%s" % ("f64" if fp64 else "f32", spp, a[1], a[3], b[1], b[3], W * H * spp / b[1] / 1e3, same), flush=True)
