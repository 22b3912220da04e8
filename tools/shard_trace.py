"""Developer script (GPU box, under rocprofv3 --kernel-trace --stats): one eighth of configuration 3 (shard 3 of 8, list scan) rendered five times -
what the launch's kernels take when the frame is split 8 ways."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 1200, 800, 500
s = rrt_amd.Scene(scene_path("final"), W, H)
r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, shard_rank=3, shard_count=8, tile_rows=4, collect_stats=False)
r.render(s)
for _ in range(5):
    r.render()
print("kernel_ms", r.stats["kernel_ms"])
r.close()
