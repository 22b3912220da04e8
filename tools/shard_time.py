"""Developer script (GPU box): kernel time of ONE of the 8 row-tile shards of BASELINE configuration 3 (final.txt 1200x800
spp 500) - what one GPU of an 8-GPU strong-scaling run does; north_star's 7.5 x needs <= 77.5 / 7.5 = 10.3 ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 500
s = rrt_amd.Scene(scene_path("final"), W, H)
for bvh in (False, True):
    full = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh); full.render(s)
    tf = min((full.render(), full.stats["kernel_ms"])[1] for _ in range(3)); full.close()
    ts = []
    for rank in range(8):
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, shard_rank=rank, shard_count=8, tile_rows=4); r.render(s)
        ts.append(min((r.render(), r.stats["kernel_ms"])[1] for _ in range(3))); r.close()
    print("use_bvh=%d  full frame %.3f ms  shards %s  slowest %.3f ms -> 8-GPU kernel-only speed-up %.2f x" % (bvh, tf, ["%.2f" % t for t in ts], max(ts), tf / max(ts)), flush=True)
