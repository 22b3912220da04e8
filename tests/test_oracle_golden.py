"""The oracle (oracle/rrt_oracle.cpp) against the committed outputs of the compiled reference.

tests/golden/* was produced by tools/make_golden.py from oracle/_ref (the reference's own sources,
compiled unchanged with the product's RNG hooked in).  These tests run anywhere (no reference, no
GPU needed) and are what pins the oracle: bit equality, both precisions.
"""
import os

import numpy as np
import pytest

from _oracle import GOLDEN, Oracle, scene_path

SCENES = {"test1": scene_path("test1"), "test2": scene_path("test2"), "test3": scene_path("test3"), "final": scene_path("final"), "xform": os.path.join(GOLDEN, "scenes", "xform.txt")}
W, H, SPP = 32, 20, 3


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_parser_tables_match_reference(name, fp64):
    g = np.load(os.path.join(GOLDEN, "tables_%s_%s.npz" % (name, "f64" if fp64 else "f32")))
    t = Oracle(SCENES[name], W, H, fp64).tables()
    assert t.counts == list(g["counts"])
    for key in ("cam", "materials", "spheres", "msph", "tris"):
        assert np.array_equal(getattr(t, key), g[key]), key


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_radiance_bit_exact_vs_reference(name, fp64):
    g = np.load(os.path.join(GOLDEN, "radiance_%s_%s.npy" % (name, "f64" if fp64 else "f32")))
    fb, st = Oracle(SCENES[name], W, H, fp64).render(SPP, 50, 1984, order=0)
    assert fb.dtype == g.dtype
    assert np.array_equal(fb, g)
    assert st["samples"] == W * H * SPP


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_depth_limited_radiance(fp64):
    g = np.load(os.path.join(GOLDEN, "radiance_final_d3_%s.npy" % ("f64" if fp64 else "f32")))
    fb, _ = Oracle(SCENES["final"], 24, 16, fp64).render(8, 3, 7, order=0)
    assert np.array_equal(fb, g)


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_iterative_order_within_rounding_of_recursive(fp64):
    # rrt.cu multiplies attenuations front to back, rrt.cpp back to front: the two differ by at most
    # a few roundings of the product.  Tolerance: depth * epsilon.
    o = Oracle(SCENES["final"], W, H, fp64)
    rec, _ = o.render(SPP, 50, 1984, order=0)
    it, st = o.render(SPP, 50, 1984, order=1)
    eps = np.finfo(rec.dtype).eps
    assert np.all(np.abs(it.astype(np.float64) - rec) <= 50 * eps * np.abs(rec) + 1e-300)
    assert st["prim_tests"] == st["segments"] * 488


def test_chunked_sum_is_a_reordering_only():
    o = Oracle(SCENES["test1"], W, H, False)
    a, _ = o.render(8, 50, 1984, order=1, chunk=0)
    b, _ = o.render(8, 50, 1984, order=1, chunk=3)
    assert np.allclose(a, b, rtol=1e-6, atol=0)
    c, _ = o.render(8, 50, 1984, order=1, chunk=8)
    assert np.array_equal(a, c)


def test_quantiser_matches_reference():
    cases = np.load(os.path.join(GOLDEN, "quantise_cases.npy"))
    for row in cases:
        fp64, spp = int(row[0]), int(row[1])
        got = Oracle.convert_color(row[2:5], spp, bool(fp64))
        want = [int(v) for v in row[5:8]]
        # what reaches the image is the low byte (main.cpp:158 uint8_t(red))
        assert [g & 0xFF for g in got] == [w & 0xFF for w in want], (row, got)


def test_quantised_frames_match_reference():
    fb = np.load(os.path.join(GOLDEN, "radiance_test1_f32.npy"))
    assert np.array_equal(Oracle.quantise(fb, SPP), np.load(os.path.join(GOLDEN, "frame_test1_f32_rgb.npy")))
    fb = np.load(os.path.join(GOLDEN, "radiance_final_f64.npy"))
    assert np.array_equal(Oracle.quantise(fb, SPP), np.load(os.path.join(GOLDEN, "frame_final_f64_rgb.npy")))


def test_rng_known_answers():
    # values produced by the reference's own random_uniform() with the hooked stream
    for seed, pixel, sample, n, f32, f64 in np.load(os.path.join(GOLDEN, "rng_known_answers.npy")):
        args = (int(seed), int(pixel), int(sample), int(n))
        assert Oracle.rng_uniform(*args, fp64=False) == np.float32(f32)
        assert Oracle.rng_uniform(*args, fp64=True) == f64
        k0, k1, hi, lo = Oracle.rng_words(*args)
        assert (hi >> 8) * 2.0 ** -24 == f32
        assert ((hi << 21) | (lo >> 11)) * 2.0 ** -53 == f64


def test_rng_streams_are_uniform_and_distinct():
    u = np.array([Oracle.rng_uniform(1984, p, s, n) for p in range(40) for s in range(5) for n in range(10)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert len(np.unique(u)) > 0.99 * len(u)


def test_scene_errors_use_reference_exit_codes(tmp_path):
    def code(text):
        p = tmp_path / "s.txt"
        p.write_text(text)
        try:
            Oracle(str(p), 8, 8)
        except ValueError as e:
            return int(str(e).split()[-1])
        return 0

    cam = "camera 0 0 5 0 0 0 0 1 0 30 0.1 5\n"
    mat = "material m lambertian 0.5 0.5 0.5\n"
    assert code(cam + mat + "sphere 0 0 0 1 m\n") == 0
    assert code(mat + "sphere 0 0 0 1 m\n") == 4  # no camera
    assert code(cam + "sphere 0 0 0 1 m\n") == 4  # no materials
    assert code(cam + mat) == 4  # no objects
    assert code(cam + "material m plastic 1 1 1\n") == 3
    assert code(cam + mat + "obj_vtx 0 0 0\n") == 1
    assert code(cam + mat + "obj_beg 1 1\nobj_vtx 0 0 0\nobj_end\n") == 1
    with pytest.raises(ValueError):
        Oracle(str(tmp_path / "missing.txt"), 8, 8)


@pytest.mark.parametrize("cfg", ["c3", "c4", "c5"])
def test_rowhash_fixtures_are_the_oracles(cfg):
    """tests/golden/c3_rowhash_f32.npy, c4_rowhash_f64.npy, c5_rowhash_f32.npz (tools/make_rowhash.py: one 64-bit hash per row of
    the oracle's frame at the BASELINE configuration's full size) - spot-checked here against a live oracle render of a few of
    those rows, so that the fixtures the GPU tier compares every row with cannot drift from the oracle that made them."""
    import hashlib
    import json

    meta = json.load(open(os.path.join(GOLDEN, "rowhash_meta.json")))[cfg]
    w, h, spp, chunk, fp64 = meta["w"], meta["h"], meta["spp"], meta["chunk"], meta["fp64"]
    if cfg == "c5":
        fx = np.load(os.path.join(GOLDEN, "c5_rowhash_f32.npz"))
        rows, want = [int(fx["rows"][40])], {int(r): int(v) for r, v in zip(fx["rows"], fx["hash"])}
    else:
        table = np.load(os.path.join(GOLDEN, "c3_rowhash_f32.npy" if cfg == "c3" else "c4_rowhash_f64.npy"))
        assert table.shape == (h,) and len(set(table.tolist())) == h  # (every row of this scene differs from every other)
        rows, want = [2, 431], {j: int(table[j]) for j in range(h)}
    o = Oracle(scene_path("final"), w, h, fp64)
    for j in rows:
        fb, _ = o.render(spp, 50, 1984, order=1, chunk=chunk, rows=(j, j + 1))
        got = int.from_bytes(hashlib.blake2b(np.ascontiguousarray(fb[j]).tobytes(), digest_size=8).digest(), "little")
        assert got == want[j], (cfg, j)
