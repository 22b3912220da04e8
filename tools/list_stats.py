"""Developer script (GPU box): candidate statistics with camera-ray lists on/off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, 64
s = rrt_amd.Scene(scene_path("final"), W, H)
res = {}
for flags in (16, 0):
    r = rrt_amd.Rrt(W, H, spp, 50, flags=flags)
    r.render(s)
    st = r.stats
    res[flags] = st
    print("lists", "off" if flags else "on ", "segments", st["segments"], "candidates", st["candidates"], "cand/seg %.3f" % (st["candidates"] / st["segments"]), "kernel %.2f ms" % st["kernel_ms"])
    r.close()
samples = W * H * spp
extra = res[0]["candidates"] - res[16]["candidates"]
print("primary rays %d; list entries beyond true scan candidates per primary: %.2f" % (samples, extra / samples))
