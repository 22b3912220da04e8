import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import rrt_amd
from _oracle import scene_path
sc = rrt_amd.Scene(scene_path("final"), 300, 200)
r = rrt_amd.Rrt(300, 200, 4, 50, use_bvh=True); r.render(sc); print(r.stats)
