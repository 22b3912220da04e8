"""Developer script (GPU box): fixed overhead vs hand-off threshold."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
s = rrt_amd.Scene(scene_path("final"), W, H)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for hl, hi, flags in ((7, 8, 32), (7, 8, 0), (7, 8, 32), (7, 8, 0), (7, 4, 0), (7, 12, 0), (12, 8, 0), (4, 8, 0)):
    ys = []
    for spp in (16, 48, 200):
        r = rrt_amd.Rrt(W, H, spp, 50, flags=flags, handoff_lanes=hl, handoff_iters=hi)
        r.set_scene(s)
        for _ in range(5):
            r.render_device(buf.data_ptr(), 0)
        torch.cuda.synchronize()
        st = r.collect()
        ys.append(st["kernel_ms_sum"] / st["renders"])
        r.close()
    A = np.polyfit((16, 48, 200), ys, 1)
    print("handoff_lanes %2d iters %4d %s: %s -> %.4f ms/spp + %.3f ms" % (hl, hi, "(no small batches)" if flags == 32 else ("(tail kernel off)" if flags else ""), ["%.2f" % y for y in ys], A[0], A[1]), flush=True)
