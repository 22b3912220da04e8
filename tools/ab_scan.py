"""Developer script (GPU box): A/B of scan variants on final.txt (exact / filter x operand source)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 100
VARIANTS = [("exact", 1), ("filter scalar", 2), ("filter hybrid", 0), ("filter lds", 4)]
for fp64 in (False, True):
    s = rrt_amd.Scene(scene_path("final"), W, H, fp64=fp64)
    ref = None
    for name, flags in VARIANTS:
        r = rrt_amd.Rrt(W, H, spp if not fp64 else max(1, spp // 4), 50, fp64=fp64, flags=flags)
        r.render(s)
        fb = r.render()
        st = r.stats
        same = True if ref is None else np.array_equal(ref, fb)
        ref = fb if ref is None else ref
        print("fp64" if fp64 else "fp32", "%-14s" % name, "%.2f ms" % st["kernel_ms"], "%.1f Msamples/s" % (st["samples"] / st["kernel_ms"] / 1e3), "cand/seg %.3f" % (st["candidates"] / st["segments"]), "grid", st["grid_blocks"], "identical" if same else "DIFFERENT", flush=True)
        r.close()
