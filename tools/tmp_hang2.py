import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path, GOLDEN
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
path = scene_path(name) if name in ("final", "test1", "test2", "test3") else os.path.join(GOLDEN, "scenes", name + ".txt")
r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=True)
t=time.time(); r.render(rrt_amd.Scene(path, W, H)); print(name, "first render done in %.2f s, kernel %.3f ms, cells %d" % (time.time()-t, r.stats["kernel_ms"], r.stats["accel_cells"]), flush=True)
