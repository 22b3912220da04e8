"""Developer script (GPU box): end-of-launch parameters on one 1/8 shard of C3 (list scan and grid)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, 500
s = rrt_amd.Scene(scene_path("final"), W, H)
def t(bvh, **kw):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, shard_rank=3, shard_count=8, tile_rows=4, **kw); r.render(s)
    v = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(4)); r.close(); return v
for bvh in (False, True):
    print("use_bvh", bvh, "default %.3f" % t(bvh), flush=True)
    for taper in (200000, 800000, 1600000, 3200000, 6400000):
        print("  taper %8d: %.3f" % (taper, t(bvh, taper_samples=taper)), flush=True)
    for it in (2, 4, 12, 16, 32):
        print("  handoff_iters %2d: %.3f" % (it, t(bvh, handoff_iters=it)), flush=True)
    for ln in (3, 16, 32, 48, 64):
        print("  handoff_lanes %2d: %.3f" % (ln, t(bvh, handoff_lanes=ln)), flush=True)
    print("  no tail kernel: %.3f" % t(bvh, flags=8), flush=True)
    for taper, it in itertools.product((800000, 3200000), (2, 4, 16)):
        print("  taper %8d iters %2d: %.3f" % (taper, it, t(bvh, taper_samples=taper, handoff_iters=it)), flush=True)
