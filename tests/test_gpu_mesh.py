"""Triangle meshes through the acceleration grid, fp32 included.

fp64: gridded under a PROVEN bound on the residual of the reference's Moeller-Trumbore test (rrtx_grid.h) - images equal
the list scan's bit for bit (tests/test_gpu_parity.py).  fp32: that bound admits nothing of practical size, and `rrt`
(use_bvh, the default, as the BVH is the reference's: main.cpp:67) enters triangles into the grid under an EMPIRICAL
inflation all the same - the stated tolerance of include/rrtx.h (RRTX_FLAG_EXACT_ACCEL): a ray segment may resolve
differently from the sequential list scan where the ray grazes a triangle's plane, exactly the pairs for which the
reference's own BVH (boxes without inflation, bvh.h:167-175) differs from its own `-b` scan.  Bounded here:
  * the VERIFY build re-scans every walked segment sequentially: disagreements <= 1e-5 of the segments (measured: 0 of
    1.2 M on the 2 880- and 27 072-triangle meshes);
  * the frame differs from the list scan's (= the oracle's, bit for bit) in <= 1e-4 of its pixels;
  * against the REAL rrtc in its default BVH mode (its own mt19937 stream, fixture tests/golden/rrtc_mesh_*) the same
    statistics as for the sphere scenes: mean < 0.1 LSB, RMS < 1.8, 8x8 blocks < 1.5;
  * `-b` and RRTX_FLAG_EXACT_ACCEL keep the list scan's bits.
"""
import os

import numpy as np
import pytest

from _oracle import GOLDEN, Oracle, mesh_scene

pytestmark = pytest.mark.gpu

VERIFY, EXACT_ACCEL = 32, 64


def _render(gpu, path, w, h, spp, fp64=False, **kw):
    sc = gpu.Scene(path, w, h, fp64=fp64)
    r = gpu.Rrt(w, h, spp, 50, use_bvh=kw.pop("use_bvh", False), fp64=fp64, **kw)
    fb = r.render(sc)
    st = r.stats
    r.close()
    return fb, st


@pytest.mark.parametrize("nu,nv", [(16, 32), (48, 96)], ids=["2880_triangles", "27072_triangles"])
def test_fp32_mesh_in_the_grid_within_the_stated_tolerance(gpu, tmp_path, nu, nv):
    f, n_tri = mesh_scene(tmp_path / "mesh.txt", nu, nv)
    w, h, spp = 300, 200, 8
    scan, s0 = _render(gpu, f, w, h, spp)  # -b: the exact list scan
    assert s0["accel_cells"] == 0 and s0["accel_exact"] == 1
    grid, s1 = _render(gpu, f, w, h, spp, use_bvh=True)
    assert s1["accel_cells"] > 0 and s1["accel_exact"] == 0  # gridded, under the approximate rule
    assert s1["scanned_segments"] < s1["segments"] // 1000
    differ = (scan != grid).any(axis=2)
    assert differ.mean() <= 1e-4, int(differ.sum())
    assert abs(int(s1["segments"]) - int(s0["segments"])) <= 1e-4 * s0["segments"]
    q0, q1 = gpu.quantise(scan, spp).astype(int), gpu.quantise(grid, spp).astype(int)
    assert (q0 != q1).mean() <= 1e-4
    # every walked segment against the sequential scan (VERIFY build of the kernel)
    _, sv = _render(gpu, f, w, h, 2, use_bvh=True, flags=VERIFY)
    assert sv["list_mismatches"] <= 1e-5 * sv["segments"], (sv["list_mismatches"], sv["segments"])
    # the caller who wants the list scan's bits with use_bvh gets them (and pays O(n) per segment)
    exact, s2 = _render(gpu, f, w, h, spp, use_bvh=True, flags=EXACT_ACCEL)
    assert s2["accel_exact"] == 1 and np.array_equal(exact, scan)
    if n_tri < 5000:  # ... which are the oracle's
        o = Oracle(f, w, h, False)
        for j in (20, 101, 160):
            fo, _ = o.render(spp, 50, 1984, order=1, chunk=8, rows=(j, j + 1))
            assert np.array_equal(scan[j], fo[j]), j


def test_fp32_mesh_agrees_statistically_with_the_real_rrtc_bvh(gpu):
    """The fixture is the REAL rrtc (reference build, its own RNG) in its default BVH mode on tests/golden/scenes/mesh.txt
    (672 triangles, 3 instances of a UV sphere: lambertian, metal, glass) at 60x40 spp 512."""
    f = os.path.join(GOLDEN, "scenes", "mesh.txt")
    ref = np.load(os.path.join(GOLDEN, "rrtc_mesh_60x40_s512.npy")).astype(np.float64)
    for kw in (dict(use_bvh=True), dict(use_bvh=False)):
        fb, st = _render(gpu, f, 60, 40, 512, **kw)
        if kw["use_bvh"]:
            assert st["accel_cells"] > 0 and st["accel_exact"] == 0
        d = gpu.quantise(fb, 512).astype(np.float64) - ref
        assert np.all(np.abs(d.mean(axis=(0, 1))) < 0.1), d.mean(axis=(0, 1))
        assert np.sqrt((d ** 2).mean()) < 1.8, np.sqrt((d ** 2).mean())
        blocks = d[:40, :56].reshape(5, 8, 7, 8, 3).mean(axis=(1, 3))
        assert np.abs(blocks).max() < 1.5, np.abs(blocks).max()


def test_a_mesh_beyond_65535_primitives_is_gridded(gpu, tmp_path):
    # cell lists hold 32-bit primitive indices: 3 x 32 256 triangles + 101 spheres
    f, n_tri = mesh_scene(tmp_path / "big.txt", 64, 256)
    assert n_tri > 90000
    w, h, spp = 96, 64, 2
    for fp64 in (True, False):
        scan, s0 = _render(gpu, f, w, h, spp, fp64=fp64)
        grid, s1 = _render(gpu, f, w, h, spp, fp64=fp64, use_bvh=True)
        assert s1["accel_cells"] > 0 and s1["accel_exact"] == (1 if fp64 else 0)
        if fp64:
            assert np.array_equal(grid, scan)
        else:
            assert ((scan != grid).any(axis=2)).mean() <= 1e-3


def _plane_scene(path, n, cam):
    """A tessellated plane (2 n^2 coplanar triangles over [-10, 10]^2 at y = 0) with 40 small spheres on it."""
    rng = np.random.default_rng(4)
    lines = [cam, "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.05", "material g dielectric 1.5"]
    for k in range(40):
        lines.append("sphere %r 0.25 %r 0.25 %s" % (float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)), "amg"[k % 3]))
    xs = np.linspace(-10, 10, n + 1)
    lines.append("obj_beg %d %d" % ((n + 1) ** 2, 2 * n * n))
    lines += ["obj_vtx %r 0.0 %r" % (float(xs[i]), float(xs[j])) for i in range(n + 1) for j in range(n + 1)]
    for i in range(n):
        for j in range(n):
            a, b, c, d = i * (n + 1) + j, i * (n + 1) + j + 1, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1
            lines += ["obj_tri %d %d %d" % (a, b, c), "obj_tri %d %d %d" % (b, d, c)]
    lines += ["obj_end", "obj 0 a"]
    path.write_text("\n".join(lines) + "\n")
    return str(path)


@pytest.mark.parametrize("cam", ["camera -14 0.12 0 0 0.0 0 0 1 0 30 0.0 14", "camera -12 1.2 0.3 0 0 0 0 1 0 35 0.02 12"], ids=["half_a_degree", "five_degrees"])
def test_fp32_plane_mesh_at_grazing_angles(gpu, tmp_path, cam):
    """The worst case for the approximate rule: rays that skim thousands of COPLANAR triangles, each with
    |a| = |e1 . (d x e2)| near the reference's cut of 1e-7 (triangle.h:49), camera rays and the bounces off the plane alike.
    Same bounds as for the sphere meshes (measured: 0 disagreements in 0.28 M / 0.40 M segments, 82 -> 3.3 ms)."""
    f = _plane_scene(tmp_path / "plane.txt", 60, cam)
    w, h, spp = 240, 180, 4
    scan, s0 = _render(gpu, f, w, h, spp)
    grid, s1 = _render(gpu, f, w, h, spp, use_bvh=True)
    assert s1["accel_cells"] > 0 and s1["accel_exact"] == 0
    assert ((scan != grid).any(axis=2)).mean() <= 1e-4
    _, sv = _render(gpu, f, w, h, 2, use_bvh=True, flags=VERIFY)
    assert sv["list_mismatches"] <= 1e-5 * sv["segments"], (sv["list_mismatches"], sv["segments"])


def _random_mesh_scene(path, seed):
    """A mesh that is NOT a UV sphere: a bumpy height field (ridges, creases, slivers where the grid is sheared), a cloud of
    free triangles of random size and orientation (1e-2 ... 0.5 units, some nearly degenerate), glass and metal among the
    materials, a camera that looks along the terrain - so that camera rays and bounces skim many triangle planes."""
    rng = np.random.default_rng(seed)
    cam = "camera %r %r %r  0 0.2 0  0 1 0  %r %r %r" % (float(rng.uniform(5, 9)), float(rng.uniform(0.4, 2.5)), float(rng.uniform(-6, 6)), float(rng.uniform(25, 50)),
                                                          float(rng.choice([0.0, 0.05])), float(rng.uniform(5, 10)))
    lines = [cam, "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.1", "material g dielectric 1.5", "material r lambertian 0.2 0.7 0.3"]
    for k in range(40):
        lines.append("sphere %r 0.3 %r 0.15 %s" % (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)), "amg"[k % 3]))
    n = int(rng.integers(40, 70))
    xs = np.linspace(-6, 6, n + 1) + rng.uniform(-0.03, 0.03, n + 1)
    zs = np.linspace(-6, 6, n + 1) + rng.uniform(-0.03, 0.03, n + 1)
    fx, fz = rng.uniform(0.5, 2.0, 2)
    hgt = 0.15 * np.sin(fx * xs[:, None]) * np.cos(fz * zs[None, :]) + 0.02 * rng.standard_normal((n + 1, n + 1)) * (rng.uniform(0, 1, (n + 1, n + 1)) < 0.3)
    shear = rng.uniform(-0.15, 0.15)  # sheared columns: slivers
    lines.append("obj_beg %d %d" % ((n + 1) ** 2, 2 * n * n))
    lines += ["obj_vtx %r %r %r" % (float(xs[i] + shear * (j % 2)), float(hgt[i, j]), float(zs[j])) for i in range(n + 1) for j in range(n + 1)]
    for i in range(n):
        for j in range(n):
            a, b, c, d = i * (n + 1) + j, i * (n + 1) + j + 1, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1
            lines += ["obj_tri %d %d %d" % (a, b, c), "obj_tri %d %d %d" % (b, d, c)]
    lines += ["obj_end", "obj 0 r"]
    m = int(rng.integers(400, 900))  # the cloud
    lines.append("obj_beg %d %d" % (3 * m, m))
    for _ in range(m):
        p = rng.uniform([-5, 0.3, -5], [5, 2.5, 5])
        size = float(np.exp(rng.uniform(np.log(0.01), np.log(0.5))))
        e1, e2 = rng.standard_normal(3), rng.standard_normal(3)
        e1 *= size / np.linalg.norm(e1)
        e2 = e2 * size / np.linalg.norm(e2) if rng.random() > 0.2 else e1 * rng.uniform(0.5, 1.5) + 1e-3 * size * e2  # a fifth: slivers
        for v in (p, p + e1, p + e2):
            lines.append("obj_vtx %r %r %r" % tuple(float(x) for x in v))
    lines += ["obj_tri %d %d %d" % (3 * k, 3 * k + 1, 3 * k + 2) for k in range(m)]
    lines += ["obj_end", "obj 1 m", "obj 1 g t 0.3 0.1 -0.2 r 30 0 1 0", "obj 1 a s 1.5 1.0 1.5 t 0 0.4 0"]
    path.write_text("\n".join(lines) + "\n")
    return str(path), 2 * n * n + 3 * m


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_fp32_random_meshes_through_the_verify_build(gpu, tmp_path, seed):
    """The approximate rule on meshes built to strain it (tests/test_tri_inflation.py has the CPU search and the stated safe
    set): the VERIFY build re-scans EVERY walked segment sequentially and counts the disagreements; the stated bound is 1e-5 of
    the segments (`include/rrtx.h`, RRTX_FLAG_EXACT_ACCEL), and the frames of the two modes may differ in <= 1e-4 of the pixels."""
    f, n_tri = _random_mesh_scene(tmp_path / "rmesh.txt", seed)
    assert n_tri > 4000
    w, h, spp = 240, 160, 4
    scan, s0 = _render(gpu, f, w, h, spp)
    grid, s1 = _render(gpu, f, w, h, spp, use_bvh=True)
    assert s1["accel_cells"] > 0 and s1["accel_exact"] == 0
    assert ((scan != grid).any(axis=2)).mean() <= 1e-4
    _, sv = _render(gpu, f, w, h, spp, use_bvh=True, flags=VERIFY)
    assert sv["segments"] > 200000
    assert sv["list_mismatches"] <= 1e-5 * sv["segments"], (sv["list_mismatches"], sv["segments"])
    exact, s2 = _render(gpu, f, w, h, spp, use_bvh=True, flags=EXACT_ACCEL)
    assert s2["accel_exact"] == 1 and np.array_equal(exact, scan)
    o = Oracle(f, w, h, False)  # ... and the list scan's rows are the oracle's
    fo, _ = o.render(spp, 50, 1984, order=1, chunk=spp, rows=(40, 41))
    assert np.array_equal(scan[40], fo[40])


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_crowded_cells_through_the_dense_pairing(gpu, tmp_path, fp64):
    """Piles of 300 and 1 300 small spheres around two points, in a scene with moving spheres (so that the accelerated kernel pairs
    (ray, entry) densely): cells of more entries than one range of the batched walk holds (250: the cell takes several ranges) and
    than four ranges hold (1 000: the cell's lane tests it alone).  Spheres and moving spheres are under the proven rule: the frame
    must be the list scan's - the oracle's - bit for bit."""
    rng = np.random.default_rng(21)
    lines = ["camera 9 2.5 6 0 0.4 0 0 1 0 35 0.02 9 0.0 1.0", "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.05", "material g dielectric 1.5", "sphere 0 -1000 0 1000 a"]
    for i in range(-6, 6):
        for j in range(-6, 6):
            lines.append("sphere %r 0.2 %r 0.2 %s" % (i + 0.9 * float(rng.uniform()), j + 0.9 * float(rng.uniform()), "amg"[(i + j) % 3]))
    for n, (cx, cy, cz) in ((300, (2.3, 0.25, -1.7)), (1300, (-3.4, 0.3, 2.6))):
        for k in range(n):
            lines.append("sphere %r %r %r %r %s" % (cx + 0.05 * float(rng.uniform()), cy + 0.05 * float(rng.uniform()), cz + 0.05 * float(rng.uniform()), 0.15 + 0.1 * float(rng.uniform()), "amg"[k % 3]))
    for k in range(6):
        x, z = float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4))
        lines.append("msphere %r 0.3 %r %r 0.5 %r 0.0 1.0 0.25 m" % (x, z, x + 0.4, z - 0.3))
    f = tmp_path / "piles.txt"
    f.write_text("\n".join(lines) + "\n")
    w, h, spp = 160, 100, 4
    scan, s0 = _render(gpu, str(f), w, h, spp, fp64=fp64)
    grid, s1 = _render(gpu, str(f), w, h, spp, fp64=fp64, use_bvh=True)
    assert s1["accel_cells"] > 0 and s1["accel_exact"] == 1 and s1["walk_pairs"] > s1["segments"]  # gridded, proven rule, densely paired
    assert np.array_equal(grid, scan) and s1["segments"] == s0["segments"]
    _, sv = _render(gpu, str(f), w, h, 2, fp64=fp64, use_bvh=True, flags=VERIFY)
    assert sv["list_mismatches"] == 0
    fo, so = Oracle(str(f), w, h, fp64).render(spp, 50, 1984, order=1, chunk=spp)
    assert np.array_equal(scan, fo) and s0["segments"] == so["segments"]


def test_exact_ties_through_the_dense_pairing(gpu, tmp_path):
    """The tie rules of the sequential scan - duplicate spheres (identical t): the LATER one wins (sphere.h:46-48 accepts root ==
    t_max); a coplanar duplicate triangle must NOT replace the earlier one (triangle.h:63 is strict); any sphere beats a triangle at
    the same t - in a scene large enough for a grid and with triangles and moving spheres in it, so that the hits are folded by the
    densely pairing kernel's atomic minimum (fp32: one packed 64-bit key; fp64: minimum of t, then maximum of the rank).  fp64
    (triangles under the proven rule): the frame must be the oracle's bit for bit; fp32: the list scan's, and the grid's within the
    stated tolerance."""
    rng = np.random.default_rng(5)
    lines = ["camera 0 1.5 7 0 0.3 0 0 1 0 40 0.0 7", "material first lambertian 0.9 0.1 0.1", "material second lambertian 0.1 0.9 0.1", "material third metal 0.2 0.2 0.9 0.0",
             "material g dielectric 1.5", "sphere 0 -1000 0 1000 first"]
    for i in range(-5, 5):
        for j in range(-5, 5):
            x, z, mat = i + 0.8 * float(rng.uniform()), j + 0.8 * float(rng.uniform()), ["first", "second", "third", "g"][(i + j) % 4]
            lines.append("sphere %r 0.2 %r 0.2 %s" % (x, z, mat))
            if (i + j) % 3 == 0:  # a duplicate of another material: the later one is what a ray sees
                lines.append("sphere %r 0.2 %r 0.2 %s" % (x, z, "second" if mat != "second" else "third"))
    lines += ["msphere 1.0 0.3 2.0 1.4 0.5 1.7 0.0 1.0 0.25 third", "msphere 1.0 0.3 2.0 1.4 0.5 1.7 0.0 1.0 0.25 first"]  # (camera shutter 0 - 0: both stand still, coincident)
    # coplanar duplicate triangles low over the ground (two instances of one obj: the first instance's material must show) and a
    # small triangle whose plane is tangent to a sphere's top (sphere and triangle report the same point at the same t)
    lines += ["obj_beg 3 1", "obj_vtx -3 0.05 -3", "obj_vtx 3 0.05 -3", "obj_vtx 0 0.05 3", "obj_tri 0 1 2", "obj_end", "obj 0 second", "obj 0 third",
              "sphere 2.5 0.6 0.5 0.3 first", "obj_beg 3 1", "obj_vtx 2.2 0.9 0.2", "obj_vtx 2.8 0.9 0.2", "obj_vtx 2.5 0.9 0.9", "obj_tri 0 1 2", "obj_end", "obj 1 second"]
    f = tmp_path / "ties_dense.txt"
    f.write_text("\n".join(lines) + "\n")
    w, h, spp = 120, 80, 4
    for fp64 in (True, False):
        scan, s0 = _render(gpu, str(f), w, h, spp, fp64=fp64)
        grid, s1 = _render(gpu, str(f), w, h, spp, fp64=fp64, use_bvh=True)
        assert s1["accel_cells"] > 0 and s1["walk_pairs"] > 0
        fo, so = Oracle(str(f), w, h, fp64).render(spp, 50, 1984, order=1, chunk=spp)
        assert np.array_equal(scan, fo) and s0["segments"] == so["segments"]
        if fp64:
            assert s1["accel_exact"] == 1 and np.array_equal(grid, fo) and s1["segments"] == so["segments"]
        else:
            assert ((grid != scan).any(axis=2)).mean() <= 1e-4
        _, sv = _render(gpu, str(f), w, h, spp, fp64=fp64, use_bvh=True, flags=VERIFY)
        assert sv["list_mismatches"] <= (0 if fp64 else 1e-5 * sv["segments"])
