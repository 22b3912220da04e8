"""Round 4's additions to the product, on the GPU: the convergence-fault counter of the wave-wide steps, the struct layout
version, and that destroying a context gives every device buffer back (advisor r03: the matrix form's operand tables leaked)."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

from _oracle import Oracle, mesh_scene, scene_path

pytestmark = pytest.mark.gpu


def test_no_wave_reaches_a_wave_wide_step_short_of_lanes(gpu):
    """dense_candidates (prefix sums by DPP, the read of lane 63, fourteen bpermutes) and the matrix-core scan take operands from
    every lane of the wave: they check EXEC on entry and count a fault instead of assuming convergence (rrtx_stats.convergence_faults,
    which rrtx_collect turns into an error).  Every variant that has such a step, both precisions, frames against the oracle."""
    mesh = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 12, 24)[0]
    cases = [(scene_path("final"), False, "scan_mfma"), (scene_path("final"), True, None), (mesh, True, "walk_pairs"), (scene_path("test3"), True, None)]  # (test3: four primitives - no grid, scanned also under use_bvh)
    for path, bvh, marker in cases:
        for fp64 in (False, True):
            w, h, spp = 96, 64, 8
            r = gpu.Rrt(w, h, spp, 50, use_bvh=bvh, fp64=fp64)
            fb = r.render(gpu.Scene(path, w, h, fp64=fp64))
            st = dict(r.stats)
            r.close()
            assert st["convergence_faults"] == 0
            if marker:
                assert st[marker] > 0, (path, marker)  # (the variant with the wave-wide step is the one that ran)
            if st["accel_exact"]:  # (an fp32 mesh is gridded under the stated tolerance: tests/test_gpu_mesh.py bounds that)
                want, so = Oracle(path, w, h, fp64).render(spp, 50, 1984, order=1, chunk=8)
                assert np.array_equal(fb, want) and st["segments"] == so["segments"]


def test_struct_layout_version_is_checked_by_the_binding(gpu):
    from rrt_amd import _lib

    assert _lib.lib.rrtx_abi_version() == _lib.ABI_VERSION == 4
    assert C.sizeof(_lib.Stats) % 8 == 0 and "convergence_faults" in dict(_lib.Stats._fields_)


def test_destroying_a_context_returns_its_device_memory(gpu):
    """A batch worker creates and destroys contexts by the hundred; every scene buffer - the matrix form's f16 table and its list
    of spheres kept apart among them (leaked up to round 3) - must be freed with the context."""
    import torch

    sc = gpu.Scene(scene_path("final"), 64, 48)

    def cycle(n):
        for _ in range(n):
            r = gpu.Rrt(64, 48, 2, 8, use_bvh=False)
            r.render(sc)
            assert r.stats["scan_mfma"] == 1
            r.close()

    cycle(5)  # (the runtime's own pools settle)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(200)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 4 << 20, "device memory shrank by %d bytes over 200 create / render / destroy cycles" % (free0 - free1)


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_reference_summation_order_at_the_speed_of_small_work_items(gpu, fp64):
    """sample_chunk = -1 asks for the reference's own order of summation (one running sum per pixel, rrt.cu:115).  Up to round 3 that
    was a SCHEDULE - one 500-sample work item per pixel, twice the time; since round 4 the work items stay small, every sample's
    radiance goes to its own slot and finalize_kernel forms the running sum.  Same bits as the one-item schedule
    (RRTX_FLAG_ONE_ITEM_PER_PIXEL) and as the oracle without any `chunk=`; both closest-hit modes, a shard, a hand-off that parks
    most of the launch, a mesh (the dense variants and their resume pass)."""
    f, w, h, spp = scene_path("final"), 120, 80, 40
    want, so = Oracle(f, w, h, fp64).render(spp, 50, 1984, order=1)  # (the oracle's default: the reference's order)
    sc = gpu.Scene(f, w, h, fp64=fp64)
    for kw in (dict(use_bvh=False), dict(use_bvh=True), dict(use_bvh=False, handoff_lanes=64, handoff_iters=1), dict(use_bvh=True, flags=gpu.FLAG_NO_TAIL_GRID),
               dict(use_bvh=False, flags=gpu.FLAG_NO_TAIL_GRID | gpu.FLAG_SCAN_NO_MFMA, handoff_lanes=64, handoff_iters=1)):
        r = gpu.Rrt(w, h, spp, 50, fp64=fp64, sample_chunk=-1, **kw)
        fb = r.render(sc)
        st = dict(r.stats)
        r.close()
        assert st["sample_chunk"] == spp and st["segments"] == so["segments"]
        assert np.array_equal(fb, want), kw
    one = gpu.Rrt(w, h, spp, 50, fp64=fp64, sample_chunk=-1, flags=gpu.FLAG_ONE_ITEM_PER_PIXEL)
    assert np.array_equal(one.render(sc), want)
    one.close()
    # a shard of the frame (rows of tiles 1, 3, 5 ... of 2 rows)
    r = gpu.Rrt(w, h, spp, 50, fp64=fp64, sample_chunk=-1, shard_rank=1, shard_count=2, tile_rows=2)
    fb = r.render(sc)
    rows = r.shard_rows()
    r.close()
    assert np.array_equal(fb[rows], want[rows])
    # a mesh through the densely pairing variants (fp64: the grid is exact there)
    if fp64:
        mesh = mesh_scene(os.path.join(tempfile.mkdtemp(), "mesh.txt"), 8, 16)[0]
        mw, mh, ms = 64, 48, 24
        want_m, _ = Oracle(mesh, mw, mh, True).render(ms, 50, 1984, order=1)
        r = gpu.Rrt(mw, mh, ms, 50, fp64=True, sample_chunk=-1, use_bvh=True)
        assert np.array_equal(r.render(gpu.Scene(mesh, mw, mh, fp64=True)), want_m)
        r.close()


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_sky_split_and_first_bounce_leave_the_frame_alone(gpu, fp64):
    """Round 4's two dense pre-passes of a launch, for scenes of spheres alone: the pixels with an empty camera-ray candidate list are finished by a kernel of
    their own (the sky split; off: RRTX_FLAG_NO_SKY_SPLIT), and the first bounce of every queued sample is left as a record (used by itself only for large
    use_bvh launches; forced here: RRTX_FLAG_FIRST_BOUNCE_ALWAYS).  Every combination, both closest-hit modes, chunked and reference-order summation, a
    hand-off that parks most of the launch, a depth limit of 1 (every record ends its path) and of 0 (no loop at all) - against the oracle, segments included."""
    f, w, h, spp = scene_path("final"), 120, 80, 24
    sc = gpu.Scene(f, w, h, fp64=fp64)
    orc = Oracle(f, w, h, fp64)
    want = {(d, ch): orc.render(spp, d, 1984, order=1, **({"chunk": ch} if ch else {})) for d in (50, 1, 0) for ch in (8, None)}
    combos = [0, gpu.FLAG_NO_SKY_SPLIT, gpu.FLAG_FIRST_BOUNCE_ALWAYS, gpu.FLAG_FIRST_BOUNCE_ALWAYS | gpu.FLAG_NO_SKY_SPLIT, gpu.FLAG_NO_FIRST_BOUNCE]
    for flags in combos:
        for kw in (dict(use_bvh=False), dict(use_bvh=True), dict(use_bvh=True, handoff_lanes=64, handoff_iters=1), dict(use_bvh=False, sample_chunk=-1),
                   dict(use_bvh=True, max_depth=1), dict(use_bvh=False, max_depth=0)):
            kw = dict(kw)
            depth = kw.pop("max_depth", 50)
            r = gpu.Rrt(w, h, spp, depth, fp64=fp64, flags=flags, **kw)
            fb = r.render(sc)
            fb2 = r.render()  # (the record buffer and the queue are per launch: a second launch of the same context)
            st = dict(r.stats)
            r.close()
            img, so = want[(depth, None if kw.get("sample_chunk") == -1 else 8)]
            assert np.array_equal(fb, img) and np.array_equal(fb2, img), (flags, kw, depth)
            assert st["segments"] == so["segments"] and st["convergence_faults"] == 0, (flags, kw, depth)
            # (the passes asked for are the passes that ran: the pre-pass belongs to use_bvh launches - the list scan was measured to gain nothing from it)
            assert st["first_bounce"] == int(bool(flags & gpu.FLAG_FIRST_BOUNCE_ALWAYS) and kw.get("use_bvh", False) and depth > 0), (flags, kw, depth)
            assert (st["sky_pixels"] > 0) == (not (flags & gpu.FLAG_NO_SKY_SPLIT) and depth > 0), (flags, kw, depth)
    # a shard of the frame, and a frame whose every pixel has candidates (no sky-only pixel: the plain queue)
    for flags in (0, gpu.FLAG_FIRST_BOUNCE_ALWAYS):
        r = gpu.Rrt(w, h, spp, 50, fp64=fp64, use_bvh=True, flags=flags, shard_rank=1, shard_count=3, tile_rows=4)
        fb = r.render(sc)
        rows = r.shard_rows()
        r.close()
        assert np.array_equal(fb[rows], want[(50, 8)][0][rows])


R4_CASES = int(os.environ.get("RRTX_R4_CASES", "8"))


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_random_scenes_of_spheres_through_the_dense_passes(gpu, tmp_path, fp64):
    """A campaign over random scenes of spheres alone (the scenes the sky split and the first-bounce pre-pass exist for), random frames, depth limits, summation
    orders, hand-offs and shards, the two passes on, off and forced, both closest-hit modes: frames and segment counts equal to the oracle's in every bit.
    RRTX_R4_CASES scales it (round 4 ran 3 000 per precision on the final kernels: no pixel off)."""
    from _oracle import crowded_scene

    rng = np.random.default_rng(404 + (1 if fp64 else 0))
    flag_sets = [0, gpu.FLAG_FIRST_BOUNCE_ALWAYS, gpu.FLAG_FIRST_BOUNCE_ALWAYS | gpu.FLAG_NO_SKY_SPLIT, gpu.FLAG_NO_SKY_SPLIT, gpu.FLAG_FIRST_BOUNCE_ALWAYS | gpu.FLAG_NO_TAIL_GRID]
    bad = []
    for case in range(R4_CASES):
        f = str(tmp_path / ("r4_%d.txt" % case))
        crowded_scene(rng, f, spheres_only=True)
        w, h = int(rng.integers(16, 129)), int(rng.integers(16, 81))
        spp = int(rng.choice([1, 3, 8, 9, 16, 24, 40]))
        depth = int(rng.choice([50, 50, 7, 2, 1]))
        reference_order = bool(rng.integers(0, 4) == 0)
        shards = int(rng.choice([1, 1, 2, 3]))
        kw = dict(tile_rows=int(rng.choice([1, 4, 16])), handoff_lanes=int(rng.choice([0, 0, 12, 64])), handoff_iters=int(rng.choice([0, 1, 12])), flags=int(rng.choice(flag_sets)))
        if reference_order:
            kw["sample_chunk"] = -1
        want, so = Oracle(f, w, h, fp64).render(spp, depth, 1984, order=1, **({} if reference_order else {"chunk": 8}))
        sc = gpu.Scene(f, w, h, fp64=fp64)
        for use_bvh in (False, True):
            got = np.zeros_like(want)
            segments = 0
            for rank in range(shards):
                r = gpu.Rrt(w, h, spp, depth, use_bvh=use_bvh, fp64=fp64, shard_rank=rank, shard_count=shards, **kw)
                part = r.render(sc)
                rows = r.shard_rows()
                got[rows] = part[rows]
                st = dict(r.stats)
                segments += st["segments"]
                r.close()
            if not (np.array_equal(got, want) and segments == so["segments"]):
                bad.append((case, "use_bvh" if use_bvh else "list scan", w, h, spp, depth, shards, kw, float((got != want).any(axis=2).mean()), segments, so["segments"]))
        if case % 50 == 49:
            print("round-4 campaign %s: %d cases, %d bad" % ("f64" if fp64 else "f32", case + 1, len(bad)), flush=True)
    assert not bad, bad
