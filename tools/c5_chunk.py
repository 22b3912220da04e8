"""Developer script (GPU box): BASELINE configuration 5 (3840x2160 spp 1000) - samples per task against kernel time, full
frame and one of its 8 row-tile shards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 3840, 2160, 1000
s = rrt_amd.Scene(scene_path("final"), W, H)
def t(bvh, shard, **kw):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, shard_rank=3 if shard else 0, shard_count=8 if shard else 1, tile_rows=4, **kw); r.render(s)
    v = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(2)); c = r.stats["sample_chunk"]; r.close(); return v, c
for bvh in (False, True):
    for chunk in [int(x) for x in sys.argv[1:]] or [0, 16, 32, 64]:
        (tf, c), (ts, _) = t(bvh, False, sample_chunk=chunk), t(bvh, True, sample_chunk=chunk)
        print("use_bvh %d chunk %3d: full %.2f ms  shard %.2f ms  (8 x shard / full = %.3f)" % (bvh, c, tf, ts, 8 * ts / tf), flush=True)
