"""Developer script (GPU box): the output side of a 4K frame (BASELINE configuration 5's size) against its render:
rrtx_quantise, rrtx_write_png, rrtx_write_ppm (P3 text, ~90 MB) - and the whole `rrt` process writing PNG / PPM."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from _oracle import scene_path
W, H, spp = 3840, 2160, 20
r = rrt_amd.Rrt(W, H, spp, 50); fb = r.render(rrt_amd.Scene(scene_path("final"), W, H)); print("render kernel %.1f ms at spp %d (spp 1000: ~650 ms)" % (r.stats["kernel_ms"], spp)); r.close()
def best(f, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3
rgb = rrt_amd.quantise(fb, spp)
print("quantise %.1f ms" % best(lambda: rrt_amd.quantise(fb, spp)))
print("write_png %.1f ms (%d bytes)" % (best(lambda: rrt_amd.write_png("/tmp/x.png", rgb)), os.path.getsize("/tmp/x.png")))
print("write_ppm %.1f ms (%d bytes)" % (best(lambda: rrt_amd.write_ppm("/tmp/x.ppm", rgb), 3), os.path.getsize("/tmp/x.ppm")), flush=True)
exe = os.path.join(ROOT, "rrt")
for args, what in ((["-o", "/tmp/y.png"], "PNG"), ([], "PPM to a file on stdout")):
    t = time.perf_counter()
    with open("/tmp/y.ppm", "wb") as out:
        p = subprocess.run([exe, "-i", scene_path("final"), "-w", str(W), "-h", str(H), "-s", str(spp)] + args, stdout=out, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t
    took = [l for l in p.stderr.decode().splitlines() if l.startswith("took")]
    print("rrt process, %s: %.3f s wall; %s" % (what, dt, took[0] if took else ""), flush=True)
