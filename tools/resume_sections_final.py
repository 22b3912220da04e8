"""Developer script (GPU box, -DRRTX_SECTION_DIAG -DRRTX_SECTION_RESUME build over rrt_amd/librrtx.so): where the waves of the RESUME pass that ends a
list-scan launch of final.txt spend their cycles, by section of the loop (sphere scenes only: that build hangs on meshes, EXPERIMENTS.md I 1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import scene_path
NAMES = ["hand-out / polling / hand-off", "camera rays", "LIST passes", "scan phase 1", "scan phase 2", "shading", "sample / task bookkeeping", "grid walk"]
W, H = 1200, 800
for label, kw, spp in (("C3 whole", {}, 500), ("C3 shard 3 of 8", dict(shard_rank=3, shard_count=8), 500)):
    sc = rrt_amd.Scene(scene_path("final"), W, H)
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, **kw); r.render(sc); r.render()
    out = (C.c_ulonglong * 8)(); lib.rrtx_section_diag(r._ctx, out)
    tot = sum(out)
    print("%s: kernel %.3f ms, resume pass %.2f wave-ms in all" % (label, r.stats["kernel_ms"], tot / 2.3e6))
    for k in range(8):
        print("   %-34s %5.1f %%" % (NAMES[k], 100.0 * out[k] / max(1, tot)), flush=True)
    r.close()
