"""Developer script (GPU box): LIST passes per SCAN pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
s = rrt_amd.Scene(scene_path("final"), W, H)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for lp in (-1, 1, 2, 3, 4, 6, 2):
    out = []
    for spp in (48, 504):
        r = rrt_amd.Rrt(W, H, spp, 50, list_passes=lp)
        r.set_scene(s)
        for _ in range(2):
            r.render_device(buf.data_ptr(), 0)
        torch.cuda.synchronize(); r.collect()
        for _ in range(4):
            r.render_device(buf.data_ptr(), 0)
        torch.cuda.synchronize()
        st = r.collect()
        out.append("spp %d: %.2f ms" % (spp, st["kernel_ms_sum"] / st["renders"]))
        r.close()
    print("list_passes %2d" % lp, " | ".join(out), "checksum %.6f" % float(buf.double().sum()), flush=True)
