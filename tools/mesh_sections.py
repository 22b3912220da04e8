"""Developer script (GPU box, -DRRTX_SECTION_DIAG build over rrt_amd/librrtx.so): the 27 072-triangle mesh scene with the render
kernel finishing every path itself (RRTX_FLAG_NO_TAIL_KERNEL: no hand-off, so that the kernel whose sections are counted holds
the whole launch, its long tail included): a wave's clock cycles by section, and what the dense pairing is fed."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from rrt_amd._lib import lib
from _oracle import mesh_scene
NAMES = ["hand-out / polling / hand-off", "camera rays", "listing (always-list, clip, cells)", "owners' fold", "decide", "shading", "sample / task bookkeeping", "dense (ray, entry) pairs"]
f, n = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 48, 96)
W, H, spp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (600, 400, 16)
for fp64 in (False, True):
    sc = rrt_amd.Scene(f, W, H, fp64=fp64)
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=True, fp64=fp64, flags=8); r.render(sc); r.render()
    out = (C.c_ulonglong * 8)(); lib.rrtx_section_diag(r._ctx, out)
    tot = sum(out)
    print("mesh %s %dx%d spp %d: kernel %.3f ms, %.1f wave-ms in all, %d segments" % ("f64" if fp64 else "f32", W, H, spp, r.stats["kernel_ms"], tot / 2.4e6, r.stats["segments"]), flush=True)
    for k in range(8):
        print("   %-34s %5.1f %%" % (NAMES[k], 100.0 * out[k] / max(1, tot)), flush=True)
    d = (C.c_ulonglong * 8)(); lib.rrtx_dense_diag(r._ctx, d)
    it = max(1, d[0])
    print("   per wave-iteration: %.2f trips, %.1f pairs, %.1f walking lanes, %.1f camera-ray lanes; %d wave-iterations, %.1f us each" % (d[1] / it, d[2] / it, d[6] / it, d[7] / it, d[0], tot / 2.4e3 / it), flush=True)
    r.close()
