"""Developer script (GPU box): the end of a launch with the matrix-core scan - hand-off parameters on the whole C3 frame, on an
eighth of it (shard 3 of 8) and on C2 (test1.txt spp 10)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
s = rrt_amd.Scene(scene_path("final"), W, H)
s2 = rrt_amd.Scene(scene_path("test1"), W, H)
def t(scene, spp, **kw):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, collect_stats=False, **kw); r.render(scene)
    v = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(4)); r.close()
    return v
for lanes in (0, 4, 12, 24):
    for iters in (0, 6, 24, 48):
        kw = dict(handoff_lanes=lanes, handoff_iters=iters)
        print("handoff_lanes %2d iters %2d:  C3 %.3f ms   C3/8 %.3f ms   C2 %.3f ms" % (lanes, iters, t(s, 500, **kw), t(s, 500, shard_rank=3, shard_count=8, **kw), t(s2, 10, **kw)), flush=True)
