"""Developer script (GPU box, RRTX_DIAG build copied over rrt_amd/librrtx.so): per-wave timeline."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from rrt_amd import _lib
from _oracle import scene_path
W, H, spp = (int(x) for x in sys.argv[1:4])
s = rrt_amd.Scene(scene_path("final"), W, H)
r = rrt_amd.Rrt(W, H, spp, 50)
r.render(s)
buf = np.zeros((131072, 8), dtype=np.uint64)
_lib.lib.rrtx_diag_read(r._ctx, buf.ctypes.data_as(C.c_void_p))
r.render()
_lib.lib.rrtx_diag_read(r._ctx, buf.ctypes.data_as(C.c_void_p))
print("kernel_ms", r.stats["kernel_ms"], "grid", r.stats["grid_blocks"])
m = buf[:65536]; m = m[m[:, 0] > 0]
t = buf[65536:]; t = t[t[:, 0] > 0]
t0 = m[:, 0].min()
us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0  # s_memrealtime ticks at 100 MHz
print("render waves", len(m), " start us: min %.1f max %.1f" % (us(m[:, 0]).min(), us(m[:, 0]).max()))
dry = m[m[:, 1] > 0]
print("queue dry seen by %d waves at us: min %.1f median %.1f max %.1f" % (len(dry), us(dry[:, 1]).min(), np.median(us(dry[:, 1])), us(dry[:, 1]).max()))
print("render exit us: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(us(m[:, 2]), [0, 10, 50, 90, 99, 100])))
print("iterations per wave: median %d max %d; iterations after dry: median %d p90 %d max %d" % (np.median(m[:, 3]), m[:, 3].max(), np.median(m[:, 4]), np.percentile(m[:, 4], 90), m[:, 4].max()))
dur = (m[:, 2] - m[:, 0]).astype(np.float64) / 100.0
print("us per iteration (whole wave life): median %.2f" % np.median(dur / m[:, 3]))
late = dry[(dry[:, 4] > 0)]
print("us per iteration after dry: median %.2f" % np.median(((late[:, 2] - late[:, 1]).astype(np.float64) / 100.0) / late[:, 4]))
if len(t):
    print("tail waves", len(t), "items", int(t[0, 4]), " start us: min %.1f  exit us: median %.1f p99 %.1f max %.1f  segments/wave: median %d max %d" % (us(t[:, 0]).min(), np.median(us(t[:, 2])), np.percentile(us(t[:, 2]), 99), us(t[:, 2]).max(), np.median(t[:, 3]), t[:, 3].max()))
    d2 = (t[:, 2] - t[:, 0]).astype(np.float64) / 100.0
    ok = t[:, 3] > 0
    print("tail us per segment: median %.2f" % np.median(d2[ok] / t[ok, 3]))
seg = r.stats["segments"]
ideal_iters = seg / 64.0 / len(m)
per_it = np.median(dur / m[:, 3])
print("segments %d -> ideal iterations per wave %.1f x %.2f us = %.1f us; lane efficiency overall %.3f" % (seg, ideal_iters, per_it, ideal_iters * per_it, seg / 64.0 / m[:, 3].sum()))
if os.environ.get("DIAG_LATE"):
    order = np.argsort(m[:, 1])[::-1][:24]
    idx = np.nonzero(buf[:65536, 0] > 0)[0]
    for o in order:
        print("wave %5d (block %4d) dry %.1f exit %.1f iters %d after-dry %d  us/iter %.2f | last pull at %.1f us, iteration %d, base %d of %d, pulls %d" % (idx[o], idx[o] // 4, us(m[o:o+1, 1])[0], us(m[o:o+1, 2])[0], m[o, 3], m[o, 4], dur[o] / m[o, 3], us(m[o:o+1, 5])[0], int(m[o, 6]) & 0xFFFFFFFF, m[o, 7], r.stats.get("total_tasks", 0), int(m[o, 6]) >> 32))
    h, edges = np.histogram(us(dry[:, 1]), bins=20)
    print("dry-time histogram:", list(zip(edges[:-1].round(0).tolist(), h.tolist())))
