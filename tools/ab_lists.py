"""Developer script (GPU box): camera-ray lists on/off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for fp64 in (False, True):
    s = rrt_amd.Scene(scene_path("final"), W, H, fp64=fp64)
    buf = torch.zeros((H, W, 3), dtype=torch.float64 if fp64 else torch.float32, device="cuda")
    for flags in (16, 0, 16, 0):
        out = []
        for spp in ((48, 504) if not fp64 else (24, 104)):
            r = rrt_amd.Rrt(W, H, spp, 50, fp64=fp64, flags=flags)
            r.set_scene(s)
            for _ in range(4):
                r.render_device(buf.data_ptr(), 0)
            torch.cuda.synchronize()
            st = r.collect()
            out.append("spp %d: %.2f ms" % (spp, st["kernel_ms_sum"] / st["renders"]))
            r.close()
        print("fp64" if fp64 else "fp32", "lists off" if flags else "lists on ", " | ".join(out), "checksum %.6f" % float(buf.double().sum()), flush=True)
