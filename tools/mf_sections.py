"""Developer script (GPU box, with a -DRRTX_SECTION_DIAG build copied over rrt_amd/librrtx.so): where a wave of the list-scan
kernel spends its cycles, filter on the vector unit against filter on the matrix cores.  args: spp"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import scene_path
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 100
NAMES = ["hand-out / polling / hand-off", "camera rays", "camera-ray lists (LIST passes)", "scan phase 1 (filter)", "scan phase 2 (exact refinement)", "shading", "sample / task bookkeeping", "-"]
for fp64 in (False, True):
    s = rrt_amd.Scene(scene_path("final"), W, H, fp64=fp64)
    for flags in (128 | 256, 128):
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=False, fp64=fp64, flags=flags)  # (128: tail kernel - the resume pass is a render_kernel too and would add its cycles)
        r.render(s)
        r.render()
        out = (C.c_ulonglong * 8)()
        lib.rrtx_section_diag(r._ctx, out)
        tot = sum(out)
        print("%s %s kernel %.3f ms" % ("f64" if fp64 else "f32", "valu" if flags & 256 else "mfma", r.stats["kernel_ms"]))
        for k in range(7):
            print("   %-34s %5.1f %%  %7.3f ms" % (NAMES[k], 100.0 * out[k] / max(1, tot), r.stats["kernel_ms"] * out[k] / max(1, tot)))
        r.close()
