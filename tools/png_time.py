"""Developer script (GPU box): wall time of the output side of a frame - rrtx_quantise and rrtx_write_png - on a frame
rendered at the reference's animation settings (1280x720 spp 50), alone and with two of them running side by side."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H = 1280, 720
for spp in (1, 50, 500):
    r = rrt_amd.Rrt(W, H, spp, 50); fb = r.render(rrt_amd.Scene(scene_path("final"), W, H)); r.close()
    n = 20
    t = time.perf_counter()
    for _ in range(n): rgb = rrt_amd.quantise(fb, spp)
    tq = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n): rrt_amd.write_png("/tmp/x.png", rgb)
    tw = (time.perf_counter() - t) / n
    def both(path):
        for _ in range(n): rrt_amd.write_png(path, rrt_amd.quantise(fb, spp))
    t = time.perf_counter()
    th = [threading.Thread(target=both, args=("/tmp/y%d.png" % k,)) for k in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
    t2 = (time.perf_counter() - t) / (2 * n)
    print("spp %3d: quantise %.2f ms, write_png %.2f ms (%d bytes, %.2f of raw); two side by side: %.2f ms per frame" % (spp, tq * 1e3, tw * 1e3, os.path.getsize("/tmp/x.png"), os.path.getsize("/tmp/x.png") / (W * H * 3.0), t2 * 1e3), flush=True)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
