"""Developer script (GPU box): time library variants / tuning parameters on final.txt.

  python tools/tune.py [lib=build/librrtx_x.so] [scene=final] [size=1200x800] [spp=8,48,504] [key=v1,v2,...]...
Every `key` is an Rrt() keyword (taper_samples, handoff_iters, handoff_lanes, list_passes, sample_chunk, flags);
the cartesian product of the value lists is timed (min of 3 launches after a warm-up) and a checksum of
the frame printed, which must not change.
"""
import itertools, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = dict(a.split("=", 1) for a in sys.argv[1:])
lib = args.pop("lib", None)
if lib:  # swap the library in before the binding loads it (the GPU box's copy of the tree is scratch)
    shutil.copy(os.path.join(ROOT, lib), os.path.join(ROOT, "rrt_amd", "librrtx.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rrt_amd
from _oracle import scene_path
W, H = (int(x) for x in args.pop("size", "1200x800").split("x"))
spps = [int(x) for x in args.pop("spp", "8,48,504").split(",")]
scene = args.pop("scene", "final")
keys = sorted(args)
s = rrt_amd.Scene(scene_path(scene), W, H)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for combo in itertools.product(*[[int(v) for v in args[k].split(",")] for k in keys]):
    kw = dict(zip(keys, combo))
    out = []
    for spp in spps:
        r = rrt_amd.Rrt(W, H, spp, 50, **kw)
        r.set_scene(s)
        r.render_device(buf.data_ptr(), 0)
        torch.cuda.synchronize(); r.collect()
        best = 1e9
        for _ in range(3):
            r.render_device(buf.data_ptr(), 0)
            torch.cuda.synchronize()
            best = min(best, r.collect()["kernel_ms"])
        out.append("spp %d: %.3f ms" % (spp, best))
        r.close()
    print(lib or "product", kw, " | ".join(out), "checksum %.6f" % float(buf.double().sum()), flush=True)
