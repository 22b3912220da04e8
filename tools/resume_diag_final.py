"""Developer script (GPU box, -DRRTX_RESUME_DIAG build over rrt_amd/librrtx.so): the longest wave of the RESUME pass that ends a list-scan launch of
final.txt (whole frame, one shard of 8) and of test1.txt (C2)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import scene_path
W, H = 1200, 800
for label, scene, kw, spp in (("C3 whole", "final", {}, 500), ("C3 shard 3 of 8", "final", dict(shard_rank=3, shard_count=8), 500), ("C2", "test1", {}, 10)):
    sc = rrt_amd.Scene(scene_path(scene), W, H)
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, **kw); r.render(sc); r.render()
    out = (C.c_ulonglong * 8)(); lib.rrtx_resume_diag(r._ctx, out)
    print("%s: kernel %.3f ms; resume pass: longest wave %d iterations, %.3f ms at 2.3 GHz (%.1f us per iteration); %d waves with work, %d iterations in all, %d segments" % (
        label, r.stats["kernel_ms"], out[0], out[1] / 2.3e6, out[1] / 2.3e3 / max(1, out[0]), out[3], out[2], out[4]), flush=True)
    r.close()
