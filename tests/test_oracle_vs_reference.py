"""The oracle against the LIVE compiled reference (oracle/_ref), where that exists.

oracle/_ref is built from /root/reference by `make -C oracle ref`; it exists in the development
container and travels to the GPU box as prebuilt files.  Elsewhere these tests skip and
test_oracle_golden.py (committed fixtures from the same build) carries the pin.
"""
import os

import numpy as np
import pytest

from _oracle import GOLDEN, Oracle, Reference, have_reference, random_scene as _random_scene, scene_path

pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built (needs /root/reference)")

SCENES = {"test1": scene_path("test1"), "test2": scene_path("test2"), "test3": scene_path("test3"), "final": scene_path("final"), "xform": os.path.join(GOLDEN, "scenes", "xform.txt")}


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_live_bit_equality(name, fp64):
    w, h, spp, depth, seed = 41, 27, 5, 50, 4242  # odd sizes, a seed the fixtures do not use
    o = Oracle(SCENES[name], w, h, fp64)
    r = Reference(SCENES[name], w, h, fp64)
    to, tr = o.tables(), r.tables()
    assert to.counts == tr.counts
    for key in ("cam", "materials", "spheres", "msph", "tris"):
        assert np.array_equal(getattr(to, key), getattr(tr, key)), key
    fo, _ = o.render(spp, depth, seed, order=0)
    assert np.array_equal(fo, r.render(spp, depth, seed))


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_shallow_depths(depth):
    o = Oracle(SCENES["final"], 30, 20, False)
    r = Reference(SCENES["final"], 30, 20, False)
    assert np.array_equal(o.render(4, depth, 1984, order=0)[0], r.render(4, depth, 1984))


def test_golden_files_are_current():
    g = np.load(os.path.join(GOLDEN, "radiance_test2_f32.npy"))
    assert np.array_equal(Reference(SCENES["test2"], 32, 20, False).render(3, 50, 1984), g)


def test_reference_table_sizes_match_the_c_abi_structs():
    # the C ABI's table structs must be layout-identical to what rrt.cu marshals (rrt.cu:217-247)
    import rrt_amd.render as rr

    for fp64 in (False, True):
        sz = Reference(SCENES["test1"], 8, 8, fp64).sizeof()
        dt = rr._table_dtypes(fp64)
        assert sz[0] == (8 if fp64 else 4)
        assert [dt["camera"].itemsize, dt["material"].itemsize, dt["sphere"].itemsize, dt["msphere"].itemsize, dt["triangle"].itemsize] == sz[1:]


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_product_parser_produces_the_reference_bytes(fp64):
    # field-by-field equality of the product's tables with the reference's raw POD bytes
    import rrt_amd
    import rrt_amd.render as rr

    dt = rr._table_dtypes(fp64)
    for name, path in SCENES.items():
        ref = Reference(path, 48, 32, fp64)
        cam, mats, sph, msph, tris = ref.raw()
        t = rrt_amd.Scene(path, 48, 32, fp64=fp64).tables()
        rcam = np.frombuffer(cam.tobytes(), dtype=dt["camera"])[0]
        for f in dt["camera"].names:
            assert np.array_equal(rcam[f], t["camera"][f]), (name, f)
        for key, raw, dkey in (("spheres", sph, "sphere"), ("moving_spheres", msph, "msphere"), ("triangles", tris, "triangle")):
            r = np.frombuffer(raw.tobytes(), dtype=dt[dkey])
            assert len(r) == len(t[key])
            for f in dt[dkey].names:
                assert np.array_equal(r[f], t[key][f]), (name, key, f)
        rm = np.frombuffer(mats.tobytes(), dtype=dt["material"])
        assert np.array_equal(rm["type"], t["materials"]["type"])
        for i, ty in enumerate(rm["type"]):
            if ty in (0, 1):
                assert np.array_equal(rm["albedo"][i], t["materials"]["albedo"][i])
            if ty == 1:
                assert rm["fuzz"][i] == t["materials"]["fuzz"][i]
            if ty == 2:
                assert rm["ref_idx"][i] == t["materials"]["ref_idx"][i]


@pytest.mark.parametrize("name", ["test1", "test2", "test3", "final"])
def test_reference_bvh_and_list_scan_agree(name):
    # the product's use_bvh mode (its acceleration grid) reproduces the list scan bit for bit, so the list
    # image is the target in both modes.  How far is the reference's own BVH from that?  With per-sample RNG streams the
    # reference's own BVH build (which draws from the stream before any sample is keyed) cannot perturb
    # the samples, so its BVH image and its list image can differ only where two primitives tie in t.
    r = Reference(SCENES[name], 40, 26, False)
    a = r.render(3, 50, 1984, bvh=False)
    b = r.render(3, 50, 1984, bvh=True)
    same = np.all(a == b, axis=2).mean()
    assert same > 0.999, same


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_parsers_agree_on_random_scenes(tmp_path, fp64):
    """Differential fuzz of the three parsers on 25 random valid scene files: the reference's scene.h (compiled), the oracle's
    restatement and the product's host_scene.cpp - tables equal value for value, the product's equal to the reference's raw
    POD bytes field by field - and the oracle renders what the reference renders on them."""
    import rrt_amd
    import rrt_amd.render as rr

    dt = rr._table_dtypes(fp64)
    rng = np.random.default_rng(20261004)
    for k in range(25):
        path = str(tmp_path / ("rand%d.txt" % k))
        _random_scene(rng, path)
        w, h = int(rng.integers(8, 64)), int(rng.integers(8, 48))
        ref, o = Reference(path, w, h, fp64), Oracle(path, w, h, fp64)
        tr, to = ref.tables(), o.tables()
        assert to.counts == tr.counts, (k, to.counts, tr.counts)
        for key in ("cam", "materials", "spheres", "msph", "tris"):
            assert np.array_equal(getattr(to, key), getattr(tr, key)), (k, key)
        cam, mats, sph, msph, tris = ref.raw()
        t = rrt_amd.Scene(path, w, h, fp64=fp64).tables()
        rcam = np.frombuffer(cam.tobytes(), dtype=dt["camera"])[0]
        for f in dt["camera"].names:
            assert np.array_equal(rcam[f], t["camera"][f]), (k, f)
        for key, raw, dkey in (("spheres", sph, "sphere"), ("moving_spheres", msph, "msphere"), ("triangles", tris, "triangle")):
            r = np.frombuffer(raw.tobytes(), dtype=dt[dkey])
            assert len(r) == len(t[key]), (k, key)
            for f in dt[dkey].names:
                assert np.array_equal(r[f], t[key][f]), (k, key, f)
        rm = np.frombuffer(mats.tobytes(), dtype=dt["material"])
        assert np.array_equal(rm["type"], t["materials"]["type"]), k
        for i, ty in enumerate(rm["type"]):
            if ty in (0, 1):
                assert np.array_equal(rm["albedo"][i], t["materials"]["albedo"][i]), (k, i)
            if ty == 1:
                assert rm["fuzz"][i] == t["materials"]["fuzz"][i], (k, i)
            if ty == 2:
                assert rm["ref_idx"][i] == t["materials"]["ref_idx"][i], (k, i)
        if k < 8:
            assert np.array_equal(o.render(2, 50, 7 + k, order=0)[0], ref.render(2, 50, 7 + k)), k


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_degenerate_primitives_like_the_reference(tmp_path, fp64):
    # radius 0, negative radius, coincident spheres, zero-area triangles, moving spheres that stand still (tests/_oracle.py)
    from _oracle import degenerate_scene

    f = degenerate_scene(tmp_path / "degenerate.txt")
    o, r = Oracle(f, 60, 40, fp64), Reference(f, 60, 40, fp64)
    to, tr = o.tables(), r.tables()
    assert to.counts == tr.counts
    for key in ("cam", "materials", "spheres", "msph", "tris"):
        assert np.array_equal(getattr(to, key), getattr(tr, key), equal_nan=True), key
    assert np.array_equal(o.render(4, 50, 1984, order=0)[0], r.render(4, 50, 1984), equal_nan=True)


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_crowded_scenes_like_the_reference(tmp_path, fp64):
    """The scenes the GPU fuzz campaign draws half of its cases from (tests/_oracle.py: crowded_scene - 40 to 600 primitives
    over four orders of magnitude of size, nested and negative-radius spheres, instanced meshes, cameras inside the crowd
    and hundreds of units away): the oracle renders what the compiled reference renders on them, bit for bit, so that a GPU
    frame equal to the oracle's there is equal to the reference's."""
    from _oracle import crowded_scene

    rng = np.random.default_rng(606)
    for k in range(10):
        f = crowded_scene(rng, tmp_path / ("crowd%d.txt" % k))
        w, h = int(rng.integers(16, 56)), int(rng.integers(12, 40))
        o, r = Oracle(f, w, h, fp64), Reference(f, w, h, fp64)
        to, tr = o.tables(), r.tables()
        assert to.counts == tr.counts, k
        for key in ("cam", "materials", "spheres", "msph", "tris"):
            assert np.array_equal(getattr(to, key), getattr(tr, key)), (k, key)
        want = r.render(2, 50, 1984 + k)
        assert np.array_equal(o.render(2, 50, 1984 + k, order=0)[0], want, equal_nan=True), k
