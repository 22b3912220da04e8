"""Regenerates tests/golden/ from the compiled reference (oracle/_ref).  Container-only: needs
/root/reference to have been compiled by `make -C oracle ref`.  The fixtures are DATA (inputs and
the reference's outputs); no reference source text is stored.

    python tools/make_golden.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from _oracle import GOLDEN, REF_DIR, Reference, have_reference, scene_path

SCENES = {"test1": scene_path("test1"), "test2": scene_path("test2"), "test3": scene_path("test3"), "final": scene_path("final"),
          "xform": os.path.join(GOLDEN, "scenes", "xform.txt")}
W, H, SPP, DEPTH, SEED = 32, 20, 3, 50, 1984


def main():
    assert have_reference(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(GOLDEN, exist_ok=True)
    # 1. same-RNG radiance images + parser tables, both precisions
    for name, path in SCENES.items():
        for fp64 in (False, True):
            tag = "%s_%s" % (name, "f64" if fp64 else "f32")
            r = Reference(path, W, H, fp64)
            t = r.tables()
            np.savez_compressed(os.path.join(GOLDEN, "tables_%s.npz" % tag), counts=np.array(t.counts), cam=t.cam, materials=t.materials, spheres=t.spheres, msph=t.msph,
                                tris=t.tris, sizeof=np.array(r.sizeof()))
            fb = r.render(SPP, DEPTH, SEED)
            np.save(os.path.join(GOLDEN, "radiance_%s.npy" % tag), fb)
            print(tag, "radiance sum", float(fb.sum()))
    # a deeper one: more samples, shallow depth limit (exercises depth exhaustion)
    for fp64 in (False, True):
        r = Reference(SCENES["final"], 24, 16, fp64)
        np.save(os.path.join(GOLDEN, "radiance_final_d3_%s.npy" % ("f64" if fp64 else "f32")), r.render(8, 3, 7))

    # 2. quantiser: edge values x spp
    vals = [0.0, 1e-30, 1e-8, 0.25, 0.5, 0.998, 0.999, 0.9990001, 1.0, 1.0000001, 2.0, 1e6, 3.4e38, float("inf"), 0.998001, 0.9980009, 0.00390625, 0.0039062, 0.06249, 0.0625,
            0.56, 0.77, 0.123456789]
    spps = [1, 3, 4, 10, 500, 1000]
    rows = []
    for fp64 in (False, True):
        r = Reference(SCENES["test1"], 4, 4, fp64)
        for spp in spps:
            for v in vals:
                for scale_by_spp in (1.0, float(spp)):
                    x = v * scale_by_spp
                    rgb = [x, x * 0.5, x * 0.25]
                    out = r.convert_color(rgb, spp)
                    rows.append([int(fp64), spp] + rgb + out)
    np.save(os.path.join(GOLDEN, "quantise_cases.npy"), np.array(rows, dtype=np.float64))

    # 3. PPM text + 8-bit frame of a reference radiance image (main.cpp:140-162, color.h)
    r = Reference(SCENES["test1"], W, H, False)
    fb = np.load(os.path.join(GOLDEN, "radiance_test1_f32.npy"))
    r.ppm(fb, SPP, os.path.join(GOLDEN, "frame_test1_f32.ppm"))
    np.save(os.path.join(GOLDEN, "frame_test1_f32_rgb.npy"), r.quantise(fb, SPP))
    r64 = Reference(SCENES["final"], W, H, True)
    fb64 = np.load(os.path.join(GOLDEN, "radiance_final_f64.npy"))
    np.save(os.path.join(GOLDEN, "frame_final_f64_rgb.npy"), r64.quantise(fb64, SPP))

    # 4. RNG known answers through the reference's own random_uniform()
    ka = []
    rf, rd = Reference(SCENES["test1"], 4, 4, False), Reference(SCENES["test1"], 4, 4, True)
    for seed, pixel, sample, n in [(1984, 0, 0, 0), (1984, 0, 0, 1), (1984, 1, 0, 0), (1984, 0, 1, 0), (1984, 959999, 499, 13), (7, 8294399, 999, 200), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 5),
                                   (1984, 123456, 78, 31)]:
        ka.append([seed, pixel, sample, n, rf.rng_probe(seed, pixel, sample, n), rd.rng_probe(seed, pixel, sample, n)])
    np.save(os.path.join(GOLDEN, "rng_known_answers.npy"), np.array(ka, dtype=np.float64))

    # 5. the real rrtc (its own mt19937 stream) for the statistical (L3) comparison
    for name, w, h, spp in [("test1", 60, 40, 1024), ("final", 60, 40, 1024), ("test2", 60, 40, 512), ("test3", 60, 40, 512)]:
        out = subprocess.run([os.path.join(REF_DIR, "rrtc"), "-i", SCENES[name], "-w", str(w), "-h", str(h), "-s", str(spp)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        tok = out.split()
        assert tok[0] == b"P3" and int(tok[1]) == w and int(tok[2]) == h
        img = np.array([int(x) for x in tok[4:]], dtype=np.uint8).reshape(h, w, 3)
        np.save(os.path.join(GOLDEN, "rrtc_%s_%dx%d_s%d.npy" % (name, w, h, spp)), img)
        print("rrtc", name, img.mean())


def mesh_fixture():
    """6. a triangle mesh (our own scene, tests/_oracle.py:mesh_scene) through the real rrtc in its DEFAULT mode, the BVH
    (main.cpp:67, bvh.h:167-175): the statistical target for `rrt` on fp32 meshes, which enters them into its grid
    under the approximate rule (include/rrtx.h, RRTX_FLAG_EXACT_ACCEL)."""
    from _oracle import mesh_scene

    f, n_tri = mesh_scene(os.path.join(GOLDEN, "scenes", "mesh.txt"), 8, 16)
    w, h, spp = 60, 40, 512
    out = subprocess.run([os.path.join(REF_DIR, "rrtc"), "-i", f, "-w", str(w), "-h", str(h), "-s", str(spp)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    tok = out.split()
    assert tok[0] == b"P3" and int(tok[1]) == w and int(tok[2]) == h
    img = np.array([int(x) for x in tok[4:]], dtype=np.uint8).reshape(h, w, 3)
    np.save(os.path.join(GOLDEN, "rrtc_mesh_%dx%d_s%d.npy" % (w, h, spp)), img)
    print("rrtc mesh (%d triangles, BVH)" % n_tri, img.mean())


if __name__ == "__main__":
    if sys.argv[1:] == ["mesh"]:
        mesh_fixture()
    else:
        main()
        mesh_fixture()
