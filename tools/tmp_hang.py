import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import mesh_scene
nu, nv, W, H, spp = (int(x) for x in sys.argv[1:6])
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), nu, nv)
r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=True)
t=time.time(); r.render(rrt_amd.Scene(f, W, H)); print(n, "triangles: first render done in %.2f s, kernel %.3f ms" % (time.time()-t, r.stats["kernel_ms"]), flush=True)
