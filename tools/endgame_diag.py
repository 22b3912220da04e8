"""Developer script (GPU box, a -DRRTX_DIAG -DRRTX_RESUME_DIAG build copied over rrt_amd/librrtx.so): where the END of a launch goes.
    python tools/endgame_diag.py <scene> <W> <H> <spp> <shard_count> <use_bvh> [Rrt keyword=value ...]
Per-wave stamps of the render kernel (start, queue seen dry, exit; iterations), the resume pass's longest wave, kernel time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from rrt_amd import _lib
from _oracle import scene_path
name, W, H, spp, count, bvh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[7:])}
s = rrt_amd.Scene(scene_path(name), W, H)
r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bool(bvh), shard_rank=min(3, count - 1), shard_count=count, tile_rows=4, **kw)
r.render(s)
buf = np.zeros((131072, 8), dtype=np.uint64)
_lib.lib.rrtx_diag_read(r._ctx, buf.ctypes.data_as(C.c_void_p))
ts = []
for _ in range(3):
    r.render(); ts.append(r.stats["kernel_ms"])
_lib.lib.rrtx_diag_read(r._ctx, buf.ctypes.data_as(C.c_void_p))
r.render()
_lib.lib.rrtx_diag_read(r._ctx, buf.ctypes.data_as(C.c_void_p))
st = r.stats
print("%s %dx%d spp %d shard 1/%d use_bvh=%d %s: kernel_ms %.3f (min of 3 before: %.3f), grid %d blocks, segments %d, chunk %d" % (name, W, H, spp, count, bvh, kw, st["kernel_ms"], min(ts), st["grid_blocks"], st["segments"], st["sample_chunk"]))
m = buf[:65536]; m = m[m[:, 0] > 0]
t0 = m[:, 0].min()
us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0  # s_memrealtime ticks at 100 MHz
dry = m[m[:, 1] > 0]
print("  render waves %d; start us max %.1f; queue seen dry by %d waves: min %.1f median %.1f max %.1f" % (len(m), us(m[:, 0]).max(), len(dry), us(dry[:, 1]).min(), np.median(us(dry[:, 1])), us(dry[:, 1]).max()))
print("  render exit us: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(us(m[:, 2]), [0, 10, 50, 90, 99, 100])))
dur = (m[:, 2] - m[:, 0]).astype(np.float64) / 100.0
print("  iterations per wave: median %d max %d; after dry: median %d max %d; us per iteration: whole life median %.2f" % (np.median(m[:, 3]), m[:, 3].max(), np.median(m[:, 4]), m[:, 4].max(), np.median(dur / m[:, 3])))
late = dry[dry[:, 4] > 0]
if len(late):
    print("  us per iteration after dry: median %.2f" % np.median(((late[:, 2] - late[:, 1]).astype(np.float64) / 100.0) / late[:, 4]))
if hasattr(_lib.lib, "rrtx_resume_diag"):
    out = (C.c_ulonglong * 8)(); _lib.lib.rrtx_resume_diag(r._ctx, out)
    print("  resume pass: longest wave %d iterations, %.1f us at 2.4 GHz (%.2f us per iteration); %d waves with work, %d iterations in all, %d segments" % (
        out[0], out[1] / 2.4e3, out[1] / 2.4e3 / max(1, out[0]), out[3], out[2], out[4]))
print("  last render exit at %.1f us of a launch of %.1f us: %.1f us after it for resume + sum + finalize + gaps" % (us(m[:, 2]).max(), st["kernel_ms"] * 1e3, st["kernel_ms"] * 1e3 - us(m[:, 2]).max()))
