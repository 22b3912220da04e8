"""Developer script (GPU box): does handing EVERY sample out as its own task (taper_samples = all) help the launches that are all tail?
The image does not depend on the taper (finalize adds a tapered pixel's samples in the chunked order), so frames must hash alike."""
import hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rrt_amd
from _oracle import scene_path, mesh_scene
mesh = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 48, 96)[0]
CASES = [("mesh f32", mesh, 600, 400, 16, False, True, {}), ("mesh f64", mesh, 600, 400, 16, True, True, {}),
         ("c2 list", scene_path("test1"), 1200, 800, 10, False, False, {}), ("c2 accel", scene_path("test1"), 1200, 800, 10, False, True, {}),
         ("c3/8 list", scene_path("final"), 1200, 800, 500, False, False, dict(shard_rank=3, shard_count=8, tile_rows=4)),
         ("c3/8 accel", scene_path("final"), 1200, 800, 500, False, True, dict(shard_rank=3, shard_count=8, tile_rows=4)),
         ("final spp8 accel", scene_path("final"), 1200, 800, 8, False, True, {})]
for name, path, W, H, spp, fp64, accel, kw in CASES:
    sc = rrt_amd.Scene(path, W, H, fp64=fp64)
    out = []
    for taper in (0, W * H * spp // 8, W * H * spp // 2, W * H * spp):
        if name.startswith("c3/8") and taper > 0:
            taper //= 8
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=accel, fp64=fp64, taper_samples=taper, **kw)
        fb = r.render(sc)
        best = 1e9
        for _ in range(3):
            fb = r.render()
            best = min(best, r.stats["kernel_ms"])
        out.append("taper %d: %.3f ms %s" % (taper, best, hashlib.blake2b(fb.tobytes(), digest_size=4).hexdigest()))
        r.close()
    print("%-18s" % name, " | ".join(out), flush=True)
