"""Developer script (GPU box): the round's measurement table for one or more library variants.

  python tools/eval_variants.py [product] [build/librrtx_x.so ...] [-- case ...]

For each library (swapped into rrt_amd/librrtx.so of the box's scratch tree, one child process each) the kernel time (HIP
events, min of 3 launches after a warm-up) and a hash of the frame for:
  c3 / c3a      final.txt 1200x800 spp 500 fp32, list scan / use_bvh
  c4 / c4a      the same in fp64
  c2 / c2a      test1.txt 1200x800 spp 10 fp32
  sh / sha      shard 3 of 8 (tile_rows 4) of c3 / c3a
  mesh / mesh64 27 072-triangle mesh 600x400 spp 16, use_bvh, fp32 / fp64
Frames must hash alike across variants (the image is part of the contract)."""
import hashlib, os, shutil, subprocess, sys, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL = ["c3", "c3a", "c4", "c4a", "c2", "c2a", "sh", "sha", "mesh", "mesh64"]


def child(cases, flags=0):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rrt_amd
    from _oracle import scene_path, mesh_scene
    mesh_file = None
    out = {}
    for c in cases:
        accel = c.endswith("a") and c not in ("mesh",)
        kw = {}
        if c in ("c3", "c3a", "c4", "c4a"):
            path, W, H, spp, fp64 = scene_path("final"), 1200, 800, 500, c.startswith("c4")
        elif c in ("c2", "c2a"):
            path, W, H, spp, fp64 = scene_path("test1"), 1200, 800, 10, False
        elif c in ("sh", "sha"):
            path, W, H, spp, fp64 = scene_path("final"), 1200, 800, 500, False
            kw = dict(shard_rank=3, shard_count=8, tile_rows=4)
        else:
            if mesh_file is None:
                mesh_file = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 48, 96)[0]
            path, W, H, spp, fp64, accel = mesh_file, 600, 400, 16, c == "mesh64", True
        sc = rrt_amd.Scene(path, W, H, fp64=fp64)
        r = rrt_amd.Rrt(W, H, spp, 50, use_bvh=accel, fp64=fp64, flags=flags, **kw)
        fb = r.render(sc)
        best = 1e9
        for _ in range(3):
            fb = r.render()
            best = min(best, r.stats["kernel_ms"])
        out[c] = (round(best, 3), hashlib.blake2b(fb.tobytes(), digest_size=6).hexdigest(), r.stats["segments"])
        r.close()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(sys.argv[3:], int(sys.argv[2]))
        sys.exit(0)
    args = sys.argv[1:]
    cases = ALL
    if "--" in args:
        k = args.index("--")
        args, cases = args[:k], args[k + 1:]
    libs = args or ["product"]
    keep = os.path.join(tempfile.mkdtemp(), "librrtx_product.so")
    shutil.copy(os.path.join(ROOT, "rrt_amd", "librrtx.so"), keep)
    table = {}
    for lib in libs:  # "path" or "path:flags" (RRTX_FLAG_* bits for every render of that row)
        path, _, fl = lib.partition(":")
        shutil.copy(keep if path == "product" else os.path.join(ROOT, path), os.path.join(ROOT, "rrt_amd", "librrtx.so"))
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", fl or "0"] + cases, capture_output=True, text=True, timeout=420)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            table[lib] = json.loads(line[0][7:]) if line else {"error": (p.stderr or p.stdout)[-600:]}
        except subprocess.TimeoutExpired:
            table[lib] = {"error": "timeout"}
            print("TIMEOUT: %s - stopping here (a hung GPU step is not followed by another)" % lib, flush=True)
            break
        finally:
            shutil.copy(keep, os.path.join(ROOT, "rrt_amd", "librrtx.so"))
        print(lib, table[lib], flush=True)
    ref = table.get(libs[0], {})
    print("\n%-34s" % "variant" + "".join("%10s" % c for c in cases))
    for lib in libs:
        t = table.get(lib, {})
        if "error" in t:
            print("%-34s ERROR %s" % (os.path.basename(lib), t["error"][-300:]))
            continue
        print("%-34s" % os.path.basename(lib) + "".join("%9.3f%s" % (t[c][0], " " if ref.get(c, [0, t[c][1]])[1] == t[c][1] else "!") if c in t else "%10s" % "-" for c in cases))
    print("('!' = the frame differs from the first variant's)")
