"""Developer script (GPU box, -DRRTX_RESUME_DIAG build over rrt_amd/librrtx.so): the resume pass's longest wave, for a scene
whose grid lives in HBM (a 27 072-triangle mesh) and one whose grid lives in LDS (final.txt).  DESIGN.md 9.4 has the numbers."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import mesh_scene, scene_path
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 48, 96)
for name, path, W, H, spp in (("mesh", f, 600, 400, 16), ("final", scene_path("final"), 1200, 800, 48)):
    for fp64 in (False, True):
        sc = rrt_amd.Scene(path, W, H, fp64=fp64)
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=True, fp64=fp64); r.render(sc); r.render()
        out = (C.c_ulonglong * 8)(); lib.rrtx_resume_diag(r._ctx, out)
        print("%s %s: kernel %.3f ms; resume pass: longest wave %d iterations, %.3f ms at 2.4 GHz (%.1f us per iteration); %d waves with work, %d iterations in all, %d segments" % (
            name, "f64" if fp64 else "f32", r.stats["kernel_ms"], out[0], out[1] / 2.4e6, out[1] / 2.4e3 / max(1, out[0]), out[3], out[2], out[4]), flush=True)
        r.close()
