"""Developer script (GPU box; for rocprofv3 --kernel-trace): BASELINE configuration 2 (test1.txt 1200x800 spp 10) a few times.
args: [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s = rrt_amd.Scene(scene_path("test1"), 1200, 800)
r = rrt_amd.Rrt(1200, 800, spp, 50, use_bvh=False)
for k in range(6):
    r.render(s if k == 0 else None)
print(r.stats["kernel_ms"], r.stats["segments"] / r.stats["samples"])
