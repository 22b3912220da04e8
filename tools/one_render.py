"""Developer script (GPU box): render final.txt once or a few times (for rocprofv3 runs).  args: W H spp [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = (int(x) for x in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2
s = rrt_amd.Scene(scene_path("final"), W, H)
r = rrt_amd.Rrt(W, H, spp, 50)
for _ in range(n):
    r.render(s)
print(r.stats)
