"""Developer script (GPU box, with a -DRRTX_SECTION_DIAG build copied over rrt_amd/librrtx.so): where a wave of the render
kernel spends its clock cycles, by section of the loop.  args: spp"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import scene_path
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 100
NAMES = ["hand-out / polling / hand-off", "camera rays", "camera-ray lists (LIST passes) | use_bvh: listing (always-list, clip, cells)", "scan phase 1 (filter) | use_bvh: owners' exact tests", "scan phase 2 (exact refinement) | use_bvh: decide", "shading", "sample / task bookkeeping", "use_bvh: dense (ray, entry) pairs"]
for fp64 in (False, True):
    s = rrt_amd.Scene(scene_path("final"), W, H, fp64=fp64)
    for bvh in (False, True):
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh, fp64=fp64, flags=128)  # (tail kernel: the resume pass is a render_kernel too and would add its cycles)
        r.render(s)
        r.render()
        out = (C.c_ulonglong * 8)()
        lib.rrtx_section_diag(r._ctx, out)
        tot = sum(out)
        print("%s use_bvh=%d kernel %.3f ms" % ("f64" if fp64 else "f32", bvh, r.stats["kernel_ms"]))
        for k in range(8):
            print("   %-34s %5.1f %%" % (NAMES[k], 100.0 * out[k] / max(1, tot)))
        if bvh and hasattr(lib, "rrtx_dense_diag"):
            d = (C.c_ulonglong * 8)()
            lib.rrtx_dense_diag(r._ctx, d)
            it = max(1, d[0])
            print("   dense pairing per wave-iteration: %.2f trips, %.1f pairs, %.2f candidates (largest list of a lane %.2f), %.4f lanes over the cap, %.1f walking lanes, %.1f camera-ray lanes; %d wave-iterations" % (
                d[1] / it, d[2] / it, d[3] / it, d[4] / it, d[5] / it, d[6] / it, d[7] / it, d[0]))
        r.close()
