"""A campaign of random scenes x random launch shapes against the oracle, bit for bit.

Every case draws a scene file - over the whole grammar (tests/_oracle.py: random_scene) or crowded enough for the
acceleration grid (crowded_scene: 40 - 600 primitives of very different sizes) -, a frame size, spp (so that the
sample chunks are whole, ragged or single), a depth limit, a shard decomposition, hand-off parameters and the pass that
finishes the parked paths, renders it through the C-ABI in the list scan and in the accelerated mode, assembles the
shards, and compares with the oracle's frame: equal in every bit (accelerated fp32 scenes whose triangles were gridded
under the approximate rule: at most one pixel in 10^4, include/rrtx.h RRTX_FLAG_EXACT_ACCEL).

RRTX_FUZZ_CASES (default 16 per precision, a few seconds) scales it: the round-2 campaigns ran up to 3000 per precision;
RRTX_FUZZ_SEED moves it elsewhere.
"""
import os

import numpy as np
import pytest

from _oracle import Oracle, crowded_scene, random_scene

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("RRTX_FUZZ_CASES", "16"))
SEED = int(os.environ.get("RRTX_FUZZ_SEED", "2024"))


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_random_scenes_random_launches(gpu, tmp_path, fp64):
    rng = np.random.default_rng(SEED + (1 if fp64 else 0))
    NO_TAIL_GRID = gpu.FLAG_NO_TAIL_GRID
    bad = []
    tally = dict(cases=0, gridded=0, approximate=0, approximate_pixels_off=0)
    for case in range(CASES):
        f = str(tmp_path / ("fuzz%d.txt" % case))
        crowded = case % 2 == 1  # (enough primitives for a grid, sizes over four orders of magnitude, cameras near and far)
        (crowded_scene if crowded else random_scene)(rng, f)
        w, h = int(rng.integers(8, 161)), int(rng.integers(8, 101))
        if crowded:
            w, h = w // 2 + 8, h // 2 + 8
        spp = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 24]))
        depth = int(rng.choice([50, 50, 50, 7, 2, 1, 0]))
        shards = int(rng.choice([1, 1, 1, 2, 3, 8]))
        kw = dict(tile_rows=int(rng.choice([1, 3, 4, 16])), handoff_lanes=int(rng.choice([0, 0, 1, 12, 64])), handoff_iters=int(rng.choice([0, 0, 1, 4, 40])),
                  flags=int(rng.choice([0, 0, NO_TAIL_GRID])), taper_samples=int(rng.choice([0, 0, 1, 5])), list_passes=int(rng.choice([0, 0, -1, 1, 3])))
        want, so = Oracle(f, w, h, fp64).render(spp, depth, 1984, order=1, chunk=8)
        sc = gpu.Scene(f, w, h, fp64=fp64)
        for use_bvh in (False, True):
            got = np.zeros_like(want)
            segments, exact = 0, True
            for rank in range(shards):
                r = gpu.Rrt(w, h, spp, depth, use_bvh=use_bvh, fp64=fp64, shard_rank=rank, shard_count=shards, **kw)
                part = r.render(sc)
                rows = r.shard_rows()
                got[rows] = part[rows]
                segments += r.stats["segments"]
                exact = exact and (not use_bvh or bool(r.stats["accel_exact"]) or r.stats["accel_cells"] == 0)
                if use_bvh and rank == 0:
                    tally["cases"] += 1
                    tally["gridded"] += r.stats["accel_cells"] > 0
                r.close()
            differ = float((got != want).any(axis=2).mean())
            if not exact:
                tally["approximate"] += 1
                tally["approximate_pixels_off"] += int((got != want).any(axis=2).sum())
            ok = differ == 0.0 if exact else differ <= 1e-4
            if exact:
                ok = ok and segments == so["segments"]
            if not ok:
                bad.append((case, "crowded" if crowded else "grammar", "use_bvh" if use_bvh else "list scan", w, h, spp, depth, shards, kw, differ, segments, so["segments"]))
    print("fuzz %s: %s" % ("f64" if fp64 else "f32", tally))
    assert not bad, bad
