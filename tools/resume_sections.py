"""Developer script (GPU box, -DRRTX_SECTION_DIAG -DRRTX_SECTION_RESUME build over rrt_amd/librrtx.so): where the waves of
the RESUME pass spend their clock cycles, by section of the loop, for the mesh scene and final.txt."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import mesh_scene, scene_path
NAMES = ["hand-out / polling / hand-off", "camera rays", "LIST passes | dense: listing (always-list, clip, cells)", "scan phase 1 | dense: owners' exact tests", "scan phase 2 | dense: decide", "shading", "sample / task bookkeeping", "grid walk | dense: the (ray, entry) pairs"]
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
        if hasattr(lib, "rrtx_dense_diag"):
            d = (C.c_ulonglong * 8)()
            lib.rrtx_dense_diag(r._ctx, d)
            it = max(1, d[0])
            print("   dense pairing per wave-iteration: %.2f trips, %.1f pairs, %.2f candidates (largest list of a lane %.2f), %.4f lanes over the cap, %.1f walking lanes, %.1f camera-ray lanes; %d wave-iterations" % (
                d[1] / it, d[2] / it, d[3] / it, d[4] / it, d[5] / it, d[6] / it, d[7] / it, d[0]))
        r.close()
