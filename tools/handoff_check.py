"""Developer script (GPU box): handoff_iters 8 against 16 on full frames and shards of C3 (both modes) and at spp 8 / 48."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
s = rrt_amd.Scene(scene_path("final"), W, H)
def t(bvh, spp, shard, it):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, shard_rank=3 if shard else 0, shard_count=8 if shard else 1, tile_rows=4, handoff_iters=it); r.render(s)
    v = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(4)); r.close(); return v
for bvh in (False, True):
    for spp, shard in ((8, False), (48, False), (500, False), (500, True), (48, True)):
        print("use_bvh %d spp %3d %s: " % (bvh, spp, "shard" if shard else "full ") + "  ".join("iters %2d: %.3f" % (it, t(bvh, spp, shard, it)) for it in (8, 12, 16, 24)), flush=True)
