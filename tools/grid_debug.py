"""Developer script (GPU box, -DRRTX_EXPERIMENTS build over rrt_amd/librrtx.so, RRTX_DEBUG_GRID=1): the grid of the mesh scene."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import mesh_scene
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 48, 96)
for fp64 in (False, True):
    sc = rrt_amd.Scene(f, 600, 400, fp64=fp64)
    r = rrt_amd.Rrt(600, 400, 16, 50, use_bvh=True, fp64=fp64); r.render(sc); r.render()
    print(fp64, r.stats, flush=True)
    r.close()
