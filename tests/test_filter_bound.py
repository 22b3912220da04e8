"""The conservative scan filter's bound, checked numerically on the CPU (no GPU needed).

Phase 1 of the HIP scan does not evaluate the reference's discriminant (sphere.h:35-40, 18 VALU ops)
but a 7-FMA expansion of disc/|d|^2 with a safety margin, and leaves the exact test to phase 2.  That is
only legal if the filter never rejects a (ray, sphere) pair the reference accepts.  This file restates
both formulas in numpy float32 (fma emulated through float64: the product of two float32 is exact
there) exactly as rrtx_kernels.hip / rrtx_api.cpp evaluate them, and searches for a counter-example
on random and on deliberately grazing configurations.  With the shipped margin constant K = 256 none may
exist; the same search finds them readily for K <= 4, which shows the search has teeth.
"""
import numpy as np
import pytest

f32 = np.float32
EPS = 2.0 ** -24


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def reference_candidate(o, d, c, r2):
    """!(discriminant < 0), unfused fp32 in the reference's order (sphere.h:35-41)."""
    oc = (o - c).astype(f32)
    a = ((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32)
    a = (a + (d[:, 2] * d[:, 2]).astype(f32)).astype(f32)
    hb = ((oc[:, 0] * d[:, 0]).astype(f32) + (oc[:, 1] * d[:, 1]).astype(f32)).astype(f32)
    hb = (hb + (oc[:, 2] * d[:, 2]).astype(f32)).astype(f32)
    q = ((oc[:, 0] * oc[:, 0]).astype(f32) + (oc[:, 1] * oc[:, 1]).astype(f32)).astype(f32)
    q = (q + (oc[:, 2] * oc[:, 2]).astype(f32)).astype(f32)
    cc = (q - r2).astype(f32)
    disc = ((hb * hb).astype(f32) - (a * cc).astype(f32)).astype(f32)
    return ~(disc < 0), a


def filter_candidate(o, d, c, r2, a, K):
    """The device filter (render_kernel, FILTER = true) and the host threshold (upload_scene)."""
    inv = (f32(1) / np.sqrt(a).astype(f32)).astype(f32)
    n = (d * inv[:, None]).astype(f32)
    s = fma(o[:, 2], n[:, 2], fma(o[:, 1], n[:, 1], (o[:, 0] * n[:, 0]).astype(f32)))
    b = (f32(2) * fma(-s[:, None].repeat(3, 1), n, o)).astype(f32)
    o2 = fma(o[:, 2], o[:, 2], fma(o[:, 1], o[:, 1], (o[:, 0] * o[:, 0]).astype(f32)))
    g = fma(np.full_like(o2, f32(K * EPS)), o2, fma(s, s, -o2))
    c64 = c.astype(np.float64)
    c2 = (c64 * c64).sum(1)
    thr64 = (c2 - r2.astype(np.float64)) - K * EPS * (c2 + r2.astype(np.float64))
    thr = thr64.astype(f32)
    thr = np.where(thr.astype(np.float64) > thr64, np.nextafter(thr, f32(-np.inf)), thr)
    thr = np.nextafter(thr, f32(-np.inf))
    u = fma(c[:, 2], n[:, 2], fma(c[:, 1], n[:, 1], (c[:, 0] * n[:, 0]).astype(f32)))
    w = fma(b[:, 2], c[:, 2], fma(b[:, 1], c[:, 1], fma(b[:, 0], c[:, 0], g)))
    return ~(fma(u, u, w) < thr)


def make_cases(rng, n, scale_o, scale_c, rmin, rmax, grazing):
    c = (rng.standard_normal((n, 3)) * scale_c).astype(f32)
    r = np.exp(rng.uniform(np.log(rmin), np.log(rmax), n)).astype(f32)
    if grazing:  # rays aimed within a few ppm of tangency: the hardest cases for a sign decision
        t = rng.standard_normal((n, 3))
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        graze = c.astype(np.float64) + t * r[:, None].astype(np.float64) * (1 + rng.uniform(-3e-6, 3e-6, (n, 1)))
        v = rng.standard_normal((n, 3))
        v -= (v * t).sum(1, keepdims=True) * t
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = (graze - v * np.exp(rng.uniform(-2, 4, (n, 1)))).astype(f32)
        d = (v * np.exp(rng.uniform(-3, 3, (n, 1)))).astype(f32)
    else:
        o = (rng.standard_normal((n, 3)) * scale_o).astype(f32)
        d = (rng.standard_normal((n, 3)) * np.exp(rng.uniform(-3, 3, (n, 1)))).astype(f32)
    return o, d, c, (r * r).astype(f32)


CONFIGS = [(10, 10, 0.05, 2, False), (10, 10, 0.05, 2, True), (10, 1000, 900, 1100, True), (1000, 1000, 0.01, 1, True), (0.01, 0.01, 1e-3, 1e-2, True), (3, 3, 1e-3, 1e3, True)]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(c) for c in CONFIGS])
def test_no_false_negative_with_the_shipped_margin(cfg):
    rng = np.random.default_rng(11)
    o, d, c, r2 = make_cases(rng, 300000, *cfg)
    ref, a = reference_candidate(o, d, c, r2)
    fil = filter_candidate(o, d, c, r2, a, 256.0)  # kFilterK in rrtx_device.h
    assert ref.sum() > 100
    assert not np.any(ref & ~fil)


def test_the_search_finds_counter_examples_when_the_margin_is_too_small():
    rng = np.random.default_rng(11)
    misses = 0
    for cfg in CONFIGS[1:]:
        o, d, c, r2 = make_cases(rng, 200000, *cfg)
        ref, a = reference_candidate(o, d, c, r2)
        misses += int(np.sum(ref & ~filter_candidate(o, d, c, r2, a, 1.0)))
    assert misses > 100


def reference_candidate64(o, d, c, r2):
    """!(discriminant < 0) in fp64, the reference's order (rrtd)."""
    oc = o - c
    a = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
    a = a + d[:, 2] * d[:, 2]
    hb = oc[:, 0] * d[:, 0] + oc[:, 1] * d[:, 1]
    hb = hb + oc[:, 2] * d[:, 2]
    q = oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1]
    q = q + oc[:, 2] * oc[:, 2]
    disc = hb * hb - a * (q - r2)
    return ~(disc < 0)


def filter_candidate_for_fp64_rays(o64, d64, c64, r264, K):
    """make_filter_ray(Path<double>) + the fp32 table pack_scene<double> builds: everything rounded to float."""
    o, d, c = o64.astype(f32), d64.astype(f32), c64.astype(f32)
    a = fma(d[:, 2], d[:, 2], fma(d[:, 1], d[:, 1], (d[:, 0] * d[:, 0]).astype(f32)))
    inv = (f32(1) / np.sqrt(a).astype(f32)).astype(f32)
    n = (d * inv[:, None]).astype(f32)
    s = fma(o[:, 2], n[:, 2], fma(o[:, 1], n[:, 1], (o[:, 0] * n[:, 0]).astype(f32)))
    b = (f32(2) * fma(-s[:, None].repeat(3, 1), n, o)).astype(f32)
    o2 = fma(o[:, 2], o[:, 2], fma(o[:, 1], o[:, 1], (o[:, 0] * o[:, 0]).astype(f32)))
    g = fma(np.full_like(o2, f32(K * EPS)), o2, fma(s, s, -o2))
    c2 = (c64 * c64).sum(1)
    thr64 = (c2 - r264) - K * EPS * (c2 + r264)
    thr = thr64.astype(f32)
    thr = np.where(thr.astype(np.float64) > thr64, np.nextafter(thr, f32(-np.inf)), thr)
    thr = np.nextafter(thr, f32(-np.inf))
    u = fma(c[:, 2], n[:, 2], fma(c[:, 1], n[:, 1], (c[:, 0] * n[:, 0]).astype(f32)))
    w = fma(b[:, 2], c[:, 2], fma(b[:, 1], c[:, 1], fma(b[:, 0], c[:, 0], g)))
    return ~(fma(u, u, w) < thr)


def make_cases64(rng, n, scale_o, scale_c, rmin, rmax, grazing, tol):
    c = rng.standard_normal((n, 3)) * scale_c
    r = np.exp(rng.uniform(np.log(rmin), np.log(rmax), n))
    if grazing:
        t = rng.standard_normal((n, 3))
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        graze = c + t * r[:, None] * (1 + rng.uniform(-tol, tol, (n, 1)))
        v = rng.standard_normal((n, 3))
        v -= (v * t).sum(1, keepdims=True) * t
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = graze - v * np.exp(rng.uniform(-2, 4, (n, 1)))
        d = v * np.exp(rng.uniform(-3, 3, (n, 1)))
    else:
        o = rng.standard_normal((n, 3)) * scale_o
        d = rng.standard_normal((n, 3)) * np.exp(rng.uniform(-3, 3, (n, 1)))
    return o, d, c, r * r


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(c) for c in CONFIGS])
def test_fp32_filter_never_rejects_what_the_fp64_discriminant_accepts(cfg):
    """fp64 rays are filtered in fp32 too (kFilterK64 = 512 covers the rounding of ray and centres to
    float); grazing within 3 ppm, 0.1 ppm, 1 ppb and exactly tangent.  False negatives appear below ~16."""
    rng = np.random.default_rng(23)
    hits = 0
    for tol in (3e-6, 1e-7, 1e-9, 0.0):
        o, d, c, r2 = make_cases64(rng, 150000, *cfg, tol)
        ref = reference_candidate64(o, d, c, r2)
        fil = filter_candidate_for_fp64_rays(o, d, c, r2, 512.0)
        hits += int(ref.sum())
        assert not np.any(ref & ~fil), tol
    assert hits > 100
    o, d, c, r2 = make_cases64(rng, 150000, *CONFIGS[1], 1e-7)
    assert np.any(reference_candidate64(o, d, c, r2) & ~filter_candidate_for_fp64_rays(o, d, c, r2, 1.0))  # the search has teeth


def test_margin_constant_matches_the_source():
    import os
    import re

    from _oracle import ROOT

    text = open(os.path.join(ROOT, "rrt_amd", "csrc", "rrtx_device.h")).read()
    assert int(re.search(r"kFilterK\s*=\s*(\d+)", text).group(1)) == 256
    assert int(re.search(r"kFilterK64\s*=\s*(\d+)", text).group(1)) == 512
