"""Developer script (GPU box): accelerated closest hit against the list scan — images, grid-walk verification, time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from rrt_amd import _lib
from _oracle import scene_path
for fp64 in (False, True):
    for (w, h, spp) in ((300, 200, 16), (1200, 800, 48)):
        sc = rrt_amd.Scene(scene_path("final"), w, h, fp64=fp64)
        r0 = rrt_amd.Rrt(w, h, spp, 50, use_bvh=False, fp64=fp64); a = r0.render(sc); a = r0.render(sc); s0 = r0.stats; r0.close()
        r1 = rrt_amd.Rrt(w, h, spp, 50, use_bvh=True, fp64=fp64); b = r1.render(sc); b = r1.render(sc); s1 = r1.stats; r1.close()
        r2 = rrt_amd.Rrt(w, h, min(spp, 16), 50, use_bvh=True, fp64=fp64, flags=_lib.FLAG_VERIFY_LISTS); r2.render(sc); s2 = r2.stats; r2.close()
        print("fp64" if fp64 else "fp32", w, h, spp, "list %.3f ms  accel %.3f ms (cells %d, scanned segments %d of %d)  identical %s  walk mismatches %d of %d segments" % (
            s0["kernel_ms"], s1["kernel_ms"], s1["accel_cells"], s1["scanned_segments"], s1["segments"], np.array_equal(a, b), s2["list_mismatches"], s2["segments"]), flush=True)
