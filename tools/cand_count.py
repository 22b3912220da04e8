"""Developer script (GPU box): segments, scanned segments and candidates (pairs that reached the exact test) of the list scan on final.txt
per filter (flag 256: vector unit) and end-of-launch mode (128: tail kernel, 8: no hand-off) - how the idle lanes' column was found wanting."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
for (w,h,spp) in ((120,80,8),(600,400,16)):
    s = rrt_amd.Scene(scene_path("final"), w, h)
    for flags in (256, 0, 256|128, 128, 256|8, 8):
        r = rrt_amd.Rrt(w, h, spp, 50, use_bvh=False, flags=flags); r.render(s); st = r.stats; r.close()
        print(w,h,spp,"flags",flags,"mfma",st["scan_mfma"],"segments",st["segments"],"scanned",st["scanned_segments"],"candidates",st["candidates"], flush=True)
