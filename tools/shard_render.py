"""Developer script (GPU box; for rocprofv3 --kernel-trace): one shard of C3 rendered a few times.  args: use_bvh [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
bvh = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = rrt_amd.Scene(scene_path("final"), 1200, 800)
r = rrt_amd.Rrt(1200, 800, 500, 50, use_bvh=bool(bvh), shard_rank=3, shard_count=8, tile_rows=4)
for _ in range(n):
    r.render(s if _ == 0 else None)
print(r.stats["kernel_ms"])
