"""Developer script (GPU box): end of a list-scan launch through the tail kernel against the resume pass on the grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
for fp64 in (False, True):
    s = rrt_amd.Scene(scene_path("final"), W, H, fp64=fp64)
    for spp, shard in ((8, False), (48, False), (500, False), (500, True)):
        out = {}
        for flags in (128, 0):
            best = None
            for lanes in ((0,) if flags else (0, 16, 32)):
                r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, fp64=fp64, shard_rank=3 if shard else 0, shard_count=8 if shard else 1, tile_rows=4, flags=flags, handoff_lanes=lanes)
                fb = r.render(s)
                t = min((r.render(), r.stats["kernel_ms"])[1] for _ in range(4)); seg = r.stats["segments"]; r.close()
                if best is None or t < best[0]: best = (t, lanes)
                out[(flags, lanes)] = (fb, seg)
            print("%s spp %3d %s %s: %.3f ms (handoff_lanes %d)" % ("f64" if fp64 else "f32", spp, "shard" if shard else "full ", "tail kernel" if flags else "grid resume", best[0], best[1]), flush=True)
        ref = out[(128, 0)]
        print("   identical:", all(np.array_equal(v[0], ref[0]) and v[1] == ref[1] for v in out.values()), flush=True)
