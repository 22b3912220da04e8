"""Developer script (GPU box): samples per task (sample_chunk) against kernel time, full frame and 1/8 shard of C3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, 500
s = rrt_amd.Scene(scene_path("final"), W, H)
def t(bvh, shard, **kw):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, shard_rank=3 if shard else 0, shard_count=8 if shard else 1, tile_rows=4, **kw); r.render(s)
    v = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(3)); r.close(); return v
for bvh in (False, True):
    for chunk in (1, 2, 4, 8, 16, 32):
        print("use_bvh %d chunk %2d: full %.3f ms  shard %.3f ms" % (bvh, chunk, t(bvh, False, sample_chunk=chunk), t(bvh, True, sample_chunk=chunk)), flush=True)
