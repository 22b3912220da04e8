"""Developer script (GPU box): host-side cost per frame of a batch - parsing, rrtx_set_scene, render + copy-back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path
W, H, spp = 1280, 720, 50
for bvh in (True, False):
    r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=bvh)
    t = time.perf_counter(); n = 30
    for _ in range(n): s = rrt_amd.Scene(scene_path("final"), W, H)
    t_parse = (time.perf_counter() - t) / n
    r.set_scene(s)
    t = time.perf_counter()
    for _ in range(n): r.set_scene(s)
    t_set = (time.perf_counter() - t) / n
    r.render()
    t = time.perf_counter()
    for _ in range(n): r.render()
    t_render = (time.perf_counter() - t) / n
    print("use_bvh %d: parse %.3f ms, set_scene %.3f ms, render + copy-back (pageable numpy frame) %.3f ms, kernel %.3f ms" % (bvh, t_parse * 1e3, t_set * 1e3, t_render * 1e3, r.stats["kernel_ms"]), flush=True)
    r.close()
