"""Developer script (GPU box): quick parity + speed probe of the HIP path against the oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import rrt_amd
from _oracle import Oracle, scene_path


def parity():
    bad = 0
    for fp64 in (False, True):
        for sc in ["test1", "test2", "test3", "final"]:
            W, H, spp = 64, 40, 4
            s = rrt_amd.Scene(scene_path(sc), W, H, fp64=fp64)
            for chunk in (-1, 3):
                r = rrt_amd.Rrt(W, H, spp, 50, fp64=fp64, sample_chunk=chunk)
                fb = r.render(s)
                o = Oracle(scene_path(sc), W, H, fp64)
                fo, st = o.render(spp, order=1, chunk=(0 if chunk < 0 else chunk))
                nb = int((fb != fo).any(axis=2).sum())
                rel = np.abs(fb.astype(np.float64) - fo) / np.maximum(np.abs(fo), 1e-30)
                print(sc, "fp64" if fp64 else "fp32", "chunk", chunk, "mismatching pixels:", nb, "max rel", rel.max(), "segments gpu/oracle", r.stats["segments"], st["segments"], flush=True)
                bad += nb
                r.close()
    return bad


def speed(spp=50):
    W, H = 1200, 800
    s = rrt_amd.Scene(scene_path("final"), W, H)
    r = rrt_amd.Rrt(W, H, spp, 50)
    t = time.time()
    r.render(s)
    print("warm %.3fs" % (time.time() - t), r.stats, flush=True)
    r.render()
    st = r.stats
    ms = st["kernel_ms"]
    print("final 1200x800 spp%d: %.1f ms -> %.1f Msamples/s; %.3f seg/sample; algorithmic %.1f GB/s" % (spp, ms, st["samples"] / ms / 1e3, st["segments"] / st["samples"], st["bytes_algorithmic"] / ms / 1e6), flush=True)


if __name__ == "__main__":
    print(rrt_amd.query_device(0), flush=True)
    b = parity()
    speed(int(sys.argv[1]) if len(sys.argv) > 1 else 50)
    sys.exit(1 if b else 0)
