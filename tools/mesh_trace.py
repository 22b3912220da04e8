"""Developer script (GPU box; run under `rocprofv3 --kernel-trace --stats`): three renders of the 27 072-triangle mesh scene
of tests/test_gpu_mesh.py at 600x400 spp 16 with use_bvh.  args: [fp64=0|1] [nu nv]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import mesh_scene
fp64 = len(sys.argv) > 1 and sys.argv[1] == "1"
nu, nv = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (48, 96)
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), nu, nv)
r = rrt_amd.Rrt(600, 400, 16, 50, use_bvh=True, fp64=fp64)
r.render(rrt_amd.Scene(f, 600, 400, fp64=fp64))
for _ in range(3):
    r.render()
    print("%d triangles fp%d: %.3f ms, %d segments, grid cells %d" % (n, 64 if fp64 else 32, r.stats["kernel_ms"], r.stats["segments"], r.stats["accel_cells"]))
