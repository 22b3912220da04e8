"""Developer script (GPU box): hand-off parameters of a list-scan launch whose tail goes through the grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
s = rrt_amd.Scene(scene_path("final"), W, H)
def t(spp, shard, **kw):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, shard_rank=3 if shard else 0, shard_count=8 if shard else 1, tile_rows=4, **kw); r.render(s)
    v = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(4)); r.close(); return v
for spp, shard in ((8, False), (48, False), (500, True), (500, False)):
    print("spp %d %s" % (spp, "shard" if shard else "full"), flush=True)
    for it in (1, 2, 4, 8, 12):
        print("   iters %2d: " % it + "  ".join("lanes %2d: %.3f" % (ln, t(spp, shard, handoff_iters=it, handoff_lanes=ln)) for ln in (7, 32, 64)), flush=True)
