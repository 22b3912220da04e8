"""Developer script (GPU box): like tune.py, fp64.  args: [lib=...] [spp=...] key=v1,v2..."""
import itertools, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = dict(a.split("=", 1) for a in sys.argv[1:])
lib = args.pop("lib", None)
if lib:
    shutil.copy(os.path.join(ROOT, lib), os.path.join(ROOT, "rrt_amd", "librrtx.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rrt_amd
from _oracle import scene_path
W, H = 1200, 800
spps = [int(x) for x in args.pop("spp", "48,504").split(",")]
keys = sorted(args)
s = rrt_amd.Scene(scene_path("final"), W, H, fp64=True)
buf = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
for combo in itertools.product(*[[int(v) for v in args[k].split(",")] for k in keys]):
    kw = dict(zip(keys, combo))
    out = []
    for spp in spps:
        r = rrt_amd.Rrt(W, H, spp, 50, fp64=True, **kw)
        r.set_scene(s)
        r.render_device(buf.data_ptr(), 0); torch.cuda.synchronize(); r.collect()
        best = 1e9
        for _ in range(3):
            r.render_device(buf.data_ptr(), 0); torch.cuda.synchronize()
            best = min(best, r.collect()["kernel_ms"])
        out.append("spp %d: %.3f ms" % (spp, best))
        r.close()
    print(lib or "product", "fp64", kw, " | ".join(out), "checksum %.6f" % float(buf.sum()), flush=True)
