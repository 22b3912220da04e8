"""Developer script (GPU box): where does the fixed per-launch time go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
def fit(W, H, depth, spps, label, back_to_back=False):
    s = rrt_amd.Scene(scene_path("final"), W, H)
    xs, ys = [], []
    for spp in spps:
        r = rrt_amd.Rrt(W, H, spp, depth)
        r.render(s)
        if back_to_back:
            import torch
            buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            for _ in range(6):
                r.render_device(buf.data_ptr(), 0)
            torch.cuda.synchronize()
            st = r.collect()
            t = st["kernel_ms_sum"] / st["renders"]
        else:
            t = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(3))
        xs.append(spp); ys.append(t)
        r.close()
    A = np.polyfit(xs, ys, 1)
    print("%-44s %s -> %.4f ms/spp + %.3f ms fixed" % (label, ["%.2f" % y for y in ys], A[0], A[1]), flush=True)
fit(1200, 800, 50, (48, 104, 200), "1200x800 d50 isolated launches")
fit(1200, 800, 50, (48, 104, 200), "1200x800 d50 back-to-back launches", True)
fit(1200, 800, 1, (48, 104, 200), "1200x800 d1 (no long paths) isolated")
fit(1200, 800, 1, (48, 104, 200), "1200x800 d1 back-to-back", True)
fit(300, 200, 50, (48, 104, 200), "300x200 d50 isolated")
fit(300, 200, 50, (48, 104, 200), "300x200 d50 back-to-back", True)
