import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import rrt_amd
from _oracle import scene_path
s = rrt_amd.Scene(scene_path("final"), 1200, 800)
for kw in ({}, {"flags": 2}, {"flags": 4}, {"taper_samples": -1}):
    r = rrt_amd.Rrt(1200, 800, 8, 50, **kw); r.render(s); print(kw, r.stats["grid_blocks"], r.stats["kernel_ms"]); r.close()
