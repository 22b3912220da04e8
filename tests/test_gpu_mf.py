"""The list scan with its filter on the matrix cores (render_kernel LDSMODE = 3) against the oracle and against the same scan
with the filter on the vector unit (RRTX_FLAG_SCAN_NO_MFMA): frames equal in every bit, segment counts equal.

The path is taken for scenes of spheres alone whose table of f16 operands fits LDS (<= 576 spheres) and lists at most 16 spheres
apart (monomials or threshold beyond the f16 range - final.txt's r = 1000 ground -, r^2 < 1e-3); rrtx_stats.scan_mfma says whether
it was.  tests/test_filter_mfma.py holds the bound and the host's packing on the CPU.
"""
import os

import numpy as np
import pytest

from _oracle import Oracle, crowded_scene, scene_path

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("RRTX_MF_CASES", "10"))


def render_both(gpu, f, w, h, spp, depth, fp64, more_flags=0, **kw):
    sc = gpu.Scene(f, w, h, fp64=fp64)
    out = {}
    for flags in (0, gpu.FLAG_SCAN_NO_MFMA):
        r = gpu.Rrt(w, h, spp, depth, use_bvh=False, fp64=fp64, flags=flags | more_flags, **kw)
        fb = r.render(sc)
        out[flags] = (fb, dict(r.stats))
        r.close()
    return out[0], out[gpu.FLAG_SCAN_NO_MFMA]


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_final_scene_matches_the_oracle_on_the_matrix_cores(gpu, fp64):
    f = scene_path("final")
    w, h, spp = 120, 80, 8
    want, so = Oracle(f, w, h, fp64).render(spp, 50, 1984, order=1, chunk=8)
    mf, vu = render_both(gpu, f, w, h, spp, 50, fp64)
    assert mf[1]["scan_mfma"] == 1 and vu[1]["scan_mfma"] == 0 and mf[1]["scan_filter"] == 1
    assert mf[1]["block_threads"] == (768 if fp64 else 512)
    assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want)
    assert mf[1]["segments"] == so["segments"] == vu[1]["segments"]
    # the filter lets through what the exact test then rejects: somewhat more than the 7-FMA filter does (twice its margin; the ground
    # sphere, listed apart, counts for every segment).  Counted with the whole launch in the render kernel (no hand-off at its end:
    # what the resume pass tests through the grid is counted differently and varies with the timing of the launch)
    from rrt_amd._lib import FLAG_NO_TAIL_KERNEL
    mf, vu = render_both(gpu, f, w, h, spp, 50, fp64, more_flags=FLAG_NO_TAIL_KERNEL)
    assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want)
    assert 0.9 * vu[1]["candidates"] <= mf[1]["candidates"] <= 1.25 * vu[1]["candidates"]  # (which camera rays meet a LIST pass depends on the company in their wave)


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_random_scenes_of_spheres(gpu, tmp_path, fp64):
    rng = np.random.default_rng(77 + (1 if fp64 else 0))
    bad, on_mfma = [], 0
    for case in range(CASES):
        f = str(tmp_path / ("mf%d.txt" % case))
        crowded_scene(rng, f, spheres_only=True)
        w, h = int(rng.integers(16, 97)), int(rng.integers(16, 65))
        spp = int(rng.choice([1, 3, 8, 9, 16]))
        depth = int(rng.choice([50, 50, 7, 1]))
        kw = dict(handoff_lanes=int(rng.choice([0, 0, 12, 64])), list_passes=int(rng.choice([0, 0, -1, 2])), taper_samples=int(rng.choice([0, 0, 3])))
        want, so = Oracle(f, w, h, fp64).render(spp, depth, 1984, order=1, chunk=8)
        mf, vu = render_both(gpu, f, w, h, spp, depth, fp64, **kw)
        on_mfma += mf[1]["scan_mfma"]
        if not (np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"] == vu[1]["segments"]):
            bad.append((case, w, h, spp, depth, kw, mf[1]["scan_mfma"], float((mf[0] != want).any(axis=2).mean()), float((vu[0] != want).any(axis=2).mean())))
    print("matrix-core filter in use in %d of %d random scenes of spheres" % (on_mfma, CASES))
    assert not bad, bad
    assert on_mfma >= CASES // 3


def write_spheres(path, spheres, camera="camera 13 2 3 0 0 0 0 1 0 20 0.1 10"):
    lines = [camera, "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.1", "material g dielectric 1.5"]
    lines += ["sphere %r %r %r %r %s" % s for s in spheres]
    open(path, "w").write("\n".join(lines) + "\n")
    return path


def grid_of_spheres(n, rng, r=0.2):
    k = int(np.ceil(np.sqrt(n)))
    return [(float(i % k - k / 2 + 0.6 * rng.random()), r, float(i // k - k / 2 + 0.6 * rng.random()), r, "amg"[i % 3]) for i in range(n)]


@pytest.mark.parametrize("n,expect", [(1, 1), (16, 1), (17, 1), (575, 1), (576, 1), (577, 0)], ids=lambda v: str(v))
def test_table_sizes_up_to_the_limit(gpu, tmp_path, n, expect):
    """1 ... 576 spheres (36 KB of operands) go through the matrix cores, 577 do not; ragged last blocks (17, 575)."""
    rng = np.random.default_rng(n)
    f = write_spheres(str(tmp_path / "n.txt"), grid_of_spheres(n, rng))
    w, h, spp = 64, 40, 4
    want, so = Oracle(f, w, h, False).render(spp, 50, 1984, order=1, chunk=8)
    mf, vu = render_both(gpu, f, w, h, spp, 50, False)
    assert mf[1]["scan_mfma"] == expect
    assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"]


def test_spheres_the_table_cannot_hold(gpu, tmp_path):
    """Up to 16 spheres beyond the f16 operands' range are tested exactly beside the table: a huge ground, a huge sphere around the
    camera (every ray is inside it), specks with r^2 < 1e-3, spheres far out; with 17 of them the vector unit keeps the filter."""
    rng = np.random.default_rng(4)
    base = grid_of_spheres(150, rng)
    apart = [(0.0, -1000.0, 0.0, 1000.0, "a"), (0.0, 0.0, 0.0, 5000.0, "a"), (300.0, 1.0, 2.0, 1.0, "m"), (1.0, 2.0, 0.5, 0.01, "m"), (0.5, 0.3, 0.2, 0.02, "g"), (1.5, 0.25, -1.0, -0.03, "g")]
    apart += [(float(260 + 10 * i), 5.0, -30.0, 3.0, "a") for i in range(10)]
    assert len(apart) == 16
    for extra, expect in ((apart, 1), (apart + [(-400.0, 2.0, 1.0, 2.0, "a")], 0)):
        spheres = base[:70] + extra[:8] + base[70:] + extra[8:]
        f = write_spheres(str(tmp_path / ("apart%d.txt" % expect)), spheres)
        for fp64 in (False, True):
            w, h, spp = 72, 48, 5
            want, so = Oracle(f, w, h, fp64).render(spp, 50, 1984, order=1, chunk=8)
            mf, vu = render_both(gpu, f, w, h, spp, 50, fp64)
            assert mf[1]["scan_mfma"] == expect
            assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"]


def test_coincident_spheres_and_exact_ties(gpu, tmp_path):
    """Equal roots from different spheres reach a ray's owner from different lanes: the later sphere must win (sphere.h:46-48)."""
    rng = np.random.default_rng(8)
    spheres = grid_of_spheres(90, rng)
    twins = []
    for i in range(0, 60, 3):  # identical twins, hollow glass shells, a sphere inside its twin
        x, y, z, r, m = spheres[i]
        twins += [(x, y, z, r, "m"), (x, y, z, -0.9 * r, "g"), (x, y, z, r, "g")]
    f = write_spheres(str(tmp_path / "ties.txt"), [(0.0, -1000.0, 0.0, 1000.0, "a")] + spheres + twins, camera="camera 6 1.5 4 0 0.2 0 0 1 0 40 0.0 7")
    for fp64 in (False, True):
        w, h, spp = 96, 64, 6
        want, so = Oracle(f, w, h, fp64).render(spp, 50, 1984, order=1, chunk=8)
        mf, vu = render_both(gpu, f, w, h, spp, 50, fp64)
        assert mf[1]["scan_mfma"] == 1
        assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"]


def test_cameras_far_out_scale_the_operands(gpu, tmp_path):
    """|o|^2 beyond 2^14 scales the ray's operands by a power of two (and the threshold with them): cameras 150 ... 30 000 away."""
    rng = np.random.default_rng(2)
    spheres = [(0.0, -1000.0, 0.0, 1000.0, "a")] + grid_of_spheres(200, rng, r=0.35)
    for dist in (150.0, 2000.0, 30000.0):
        f = write_spheres(str(tmp_path / "far.txt"), spheres, camera="camera %r %r %r 0 0 0 0 1 0 %r 0.0 %r" % (dist * 0.8, dist * 0.5, dist * 0.33, 1200.0 / dist, dist))
        w, h, spp = 64, 48, 4
        want, so = Oracle(f, w, h, False).render(spp, 50, 1984, order=1, chunk=8)
        mf, vu = render_both(gpu, f, w, h, spp, 50, False)
        assert mf[1]["scan_mfma"] == 1
        assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"]


def test_every_ray_a_candidate_for_hundreds_of_spheres(gpu, tmp_path):
    """Nested shells around the camera and the scene: every segment's filter lets 200 spheres through - the lanes' mark lists fill
    and are emptied in mid-scan, the wave's pairs (12 800) exceed the queue that shares them out and every lane tests its own."""
    rng = np.random.default_rng(6)
    shells = [(0.1 * float(rng.standard_normal()), 1.0, 0.1 * float(rng.standard_normal()), 30.0 + 0.35 * i, "amg"[i % 3] if i > 190 else "g") for i in range(200)]
    spheres = grid_of_spheres(60, rng) + shells
    f = write_spheres(str(tmp_path / "shells.txt"), spheres, camera="camera 6 2 4 0 0.3 0 0 1 0 40 0.0 7")
    for fp64 in (False, True):
        w, h, spp = 48, 32, 3
        want, so = Oracle(f, w, h, fp64).render(spp, 50, 1984, order=1, chunk=8)
        mf, vu = render_both(gpu, f, w, h, spp, 50, fp64)
        assert mf[1]["scan_mfma"] == 1
        assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"]


def test_rays_beyond_the_operands_scale_scan_sequentially(gpu, tmp_path):
    """|o|^2 >= 2^38: the power of two that would bring the ray's operands into f16 is itself below f16's range - those lanes take
    the reference's sequential scan (camera rays; what they scatter into is near the scene again and goes through the matrix cores)."""
    rng = np.random.default_rng(12)
    spheres = [(0.0, -1000.0, 0.0, 1000.0, "a")] + grid_of_spheres(120, rng, r=0.4)
    for dist in (6.0e5, 3.0e6):
        f = write_spheres(str(tmp_path / "veryfar.txt"), spheres, camera="camera %r %r %r 0 0 0 0 1 0 %r 0.0 %r" % (dist * 0.8, dist * 0.5, dist * 0.33, 1500.0 / dist, dist))
        w, h, spp = 48, 32, 4
        for list_passes in (0, -1):  # (-1: no camera-ray lists - the camera rays themselves go through the scan)
            want, so = Oracle(f, w, h, False).render(spp, 50, 1984, order=1, chunk=8)
            mf, vu = render_both(gpu, f, w, h, spp, 50, False, list_passes=list_passes)
            assert mf[1]["scan_mfma"] == 1
            assert np.array_equal(mf[0], want) and np.array_equal(vu[0], want) and mf[1]["segments"] == so["segments"]
