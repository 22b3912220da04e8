"""Developer script (GPU box): camera-ray LIST passes allowed between two SCAN / walk passes (rrtx_params.list_passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, 500
s = rrt_amd.Scene(scene_path("final"), W, H)
for bvh in (False,):
    for lp in (-1, 1, 2, 3, 4, 6, 10):
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, list_passes=lp); r.render(s)
        t = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(3)); r.close()
        print("use_bvh %d list_passes %2d: %.3f ms" % (bvh, lp, t), flush=True)
