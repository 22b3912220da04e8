"""Developer script (GPU box, -DRRTX_SECTION_DIAG -DRRTX_SECTION_RESUME build over rrt_amd/librrtx.so): where the waves of
the RESUME pass spend their clock cycles, by section of the loop, for the mesh scene and final.txt."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import mesh_scene, scene_path
NAMES = ["hand-out / polling / hand-off", "camera rays", "camera-ray lists (LIST passes)", "scan phase 1 (filter)", "scan phase 2 (exact refinement)", "shading", "sample / task bookkeeping", "grid walk (use_bvh)"]
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 48, 96)
for name, path, W, H, spp in (("mesh", f, 600, 400, 16), ("final", scene_path("final"), 1200, 800, 48)):
    for fp64 in (False, True):
        sc = rrt_amd.Scene(path, W, H, fp64=fp64)
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=True, fp64=fp64); r.render(sc); r.render()
        out = (C.c_ulonglong * 8)(); lib.rrtx_section_diag(r._ctx, out)
        tot = sum(out)
        print("%s %s: kernel %.3f ms, resume pass %.1f wave-ms in all" % (name, "f64" if fp64 else "f32", r.stats["kernel_ms"], tot / 2.4e6))
        for k in range(8):
            print("   %-34s %5.1f %%" % (NAMES[k], 100.0 * out[k] / max(1, tot)), flush=True)
        r.close()
