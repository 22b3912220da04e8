import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import rrt_amd
from _oracle import scene_path
for (w,h,spp) in ((120,80,8),(600,400,16)):
    s = rrt_amd.Scene(scene_path("final"), w, h)
    for flags in (256, 0, 256|128, 128, 256|8, 8):
        r = rrt_amd.Rrt(w, h, spp, 50, use_bvh=False, flags=flags); r.render(s); st = r.stats; r.close()
        print(w,h,spp,"flags",flags,"mfma",st["scan_mfma"],"segments",st["segments"],"scanned",st["scanned_segments"],"candidates",st["candidates"], flush=True)
