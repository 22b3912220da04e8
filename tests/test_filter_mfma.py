"""The scan filter on the matrix cores (render_kernel LDSMODE = 3, rrtx_pack.h: pack_mf_table), checked on the CPU.

The filter's value f = (c.n)^2 + b.c + g - thr is evaluated as ONE dot product of 31 f16 x f16 terms accumulated in f32 (each f32
operand split in two f16 pieces, the ray's side scaled by a power of two).  That is only legal if a (ray, sphere) pair the
reference's discriminant accepts (sphere.h:35-41) never gets a NEGATIVE f.  This file restates the device's and the host's
arithmetic in numpy - the split, the scale, the term order of the table, an f32 accumulation in several orders and one in
float64 (the instruction's internal order and width are not documented) - and searches for a counter-example on random and on
grazing configurations.  With the shipped margin (kFilterKMf = 512 unit roundoffs, kFilterKMf64 = 1024 for fp64 rays) none may
exist; the same search finds them for K <= 1, which shows it has teeth.  The host's packing (layout, f16 rounding, the spheres
listed apart) is compiled from rrtx_pack.h and compared with the model term by term.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from test_filter_bound import EPS, f32, fma, make_cases, reference_candidate, reference_candidate64, make_cases64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f16 = np.float16
ABS_TERM = 1e-5  # pack_mf_table: what the f16 pieces lose to underflow near zero


def split64(x):
    """float64 -> (hi, lo) float16, the host's split (pack_mf_table: through float, round to nearest even)."""
    with np.errstate(over="ignore", invalid="ignore"):  # (values beyond the f16 range: their spheres are listed apart)
        hi = x.astype(f32).astype(f16)
        lo = (x - hi.astype(np.float64)).astype(f32).astype(f16)
    return hi, lo


def split32(x):
    """float32 -> (hi, lo) float16, the device's split (v_cvt_f16_f32, exact difference, v_cvt_f16_f32)."""
    hi = x.astype(f16)
    lo = (x - hi.astype(f32)).astype(f32).astype(f16)
    return hi, lo


def sphere_terms(c, r2, K):
    """-> (terms [n, 32] float16 in the table's TERM order, in_table [n] bool): pack_mf_table."""
    c64, r264 = c.astype(np.float64), r2.astype(np.float64)
    c2 = (c64 * c64).sum(1)
    v = [c64[:, 0] ** 2, c64[:, 1] ** 2, c64[:, 2] ** 2, 2 * c64[:, 0] * c64[:, 1], 2 * c64[:, 0] * c64[:, 2], 2 * c64[:, 1] * c64[:, 2], c64[:, 0], c64[:, 1], c64[:, 2]]
    thr = (c2 - r264) - K * EPS * (c2 + r264) - ABS_TERM
    ok = np.ones(len(c), bool)
    for x in v + [thr]:
        ok &= np.isfinite(x) & (np.abs(x) <= 60000.0)
    ok &= r264 >= 1e-3
    T = np.zeros((len(c), 32), f16)
    for q, x in enumerate(v):
        h, l = split64(x)
        T[:, 3 * q + 0], T[:, 3 * q + 1], T[:, 3 * q + 2] = h, h, l
    T[:, 27] = T[:, 28] = f16(1)
    T[:, 29], T[:, 30] = split64(thr)
    T[~ok] = 0
    T[~ok, 29] = f16(60000.0)
    return T, ok


def ray_terms(o, d, a, K):
    """-> terms [n, 32] float16: the render kernel's operand (make_filter_ray_mf, the power-of-two scale, the split)."""
    inv = (f32(1) / np.sqrt(a).astype(f32)).astype(f32)
    n = (d * inv[:, None]).astype(f32)
    s = fma(o[:, 2], n[:, 2], fma(o[:, 1], n[:, 1], (o[:, 0] * n[:, 0]).astype(f32)))
    b = (f32(2) * fma(-s[:, None].repeat(3, 1), n, o)).astype(f32)
    o2 = fma(o[:, 2], o[:, 2], fma(o[:, 1], o[:, 1], (o[:, 0] * o[:, 0]).astype(f32)))
    g = fma(np.full_like(o2, f32(K * EPS)), o2, fma(s, s, -o2))
    m = np.maximum(np.maximum(np.abs(g), np.abs(b[:, 0])), np.maximum(np.abs(b[:, 1]), np.abs(b[:, 2])))
    _, e = np.frexp(m)
    lam = np.ldexp(f32(1), np.where(e > 14, 14 - e, 0)).astype(f32)
    N = [n[:, 0] * n[:, 0], n[:, 1] * n[:, 1], n[:, 2] * n[:, 2], n[:, 0] * n[:, 1], n[:, 0] * n[:, 2], n[:, 1] * n[:, 2]]
    vals = [(x.astype(f32) * lam).astype(f32) for x in N] + [(b[:, i] * lam).astype(f32) for i in range(3)] + [(g * lam).astype(f32)]
    T = np.zeros((len(o), 32), f16)
    for q in range(9):
        h, l = split32(vals[q])
        T[:, 3 * q + 0], T[:, 3 * q + 1], T[:, 3 * q + 2] = h, l, h
    T[:, 27], T[:, 28] = split32(vals[9])
    T[:, 29] = T[:, 30] = (-lam).astype(f16)
    return T


def mfma_candidate(o, d, c, r2, a, K, order=None, wide=False):
    """not (f < 0) for pair i of (ray i, sphere i); spheres the table cannot hold are candidates (tested exactly)."""
    S, ok = sphere_terms(c, r2, K)
    R = ray_terms(o, d, a, K)
    P = S.astype(np.float64) * R.astype(np.float64)  # f16 x f16: exact in f32, let alone here
    if order is not None:
        P = P[:, order]
    if wide:
        acc = P.sum(1).astype(f32)
    else:
        acc = np.zeros(len(o), f32)
        for j in range(P.shape[1]):
            acc = (acc.astype(np.float64) + P[:, j]).astype(f32)
    return ~(acc < 0) | ~ok, ok


def graze_cases(rng, n, scale_o, scale_c, rmin, rmax, tol):
    c = (rng.standard_normal((n, 3)) * scale_c).astype(f32)
    r = np.exp(rng.uniform(np.log(rmin), np.log(rmax), n)).astype(f32)
    t = rng.standard_normal((n, 3))
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    graze = c.astype(np.float64) + t * r[:, None].astype(np.float64) * (1 + rng.uniform(-tol, tol, (n, 1)))
    v = rng.standard_normal((n, 3))
    v -= (v * t).sum(1, keepdims=True) * t
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = (graze - v * np.exp(rng.uniform(-2, np.log(scale_o * 3), (n, 1)))).astype(f32)
    d = (v * np.exp(rng.uniform(-3, 3, (n, 1)))).astype(f32)
    return o, d, c, (r * r).astype(f32)


K_MF, K_MF64 = 512.0, 1024.0
CONFIGS = [(10, 10, 0.05, 2, False), (10, 10, 0.05, 2, True), (3, 3, 0.04, 30, True), (100, 10, 0.05, 2, True), (1000, 10, 0.1, 1, True), (30, 100, 0.2, 5, True), (3000, 100, 0.5, 50, True)]
GRAZE = [(10, 10, 0.05, 2), (100, 10, 0.05, 2), (1000, 10, 0.1, 1), (3, 3, 0.04, 30), (30, 100, 0.2, 5), (20000, 50, 0.1, 10)]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(c) for c in CONFIGS])
def test_no_false_negative_with_the_shipped_margin(cfg):
    rng = np.random.default_rng(21)
    o, d, c, r2 = make_cases(rng, 200000, *cfg)
    ref, a = reference_candidate(o, d, c, r2)
    orders = [None, np.arange(31, -1, -1), rng.permutation(32)]
    for order in orders:
        cand, ok = mfma_candidate(o, d, c, r2, a, K_MF, order=order)
        assert int(np.sum(ref & ~cand)) == 0
    cand, ok = mfma_candidate(o, d, c, r2, a, K_MF, wide=True)
    assert int(np.sum(ref & ~cand)) == 0
    assert ok.mean() > 0.5  # (the case is about the table, not about the spheres listed apart)


@pytest.mark.parametrize("cfg", GRAZE, ids=[str(c) for c in GRAZE])
def test_no_false_negative_on_grazing_rays(cfg):
    rng = np.random.default_rng(5)
    for tol in (1e-2, 1e-4, 1e-6, 0.0):
        o, d, c, r2 = graze_cases(rng, 100000, *cfg, tol)
        ref, a = reference_candidate(o, d, c, r2)
        for kw in ({}, {"wide": True}, {"order": rng.permutation(32)}):
            cand, ok = mfma_candidate(o, d, c, r2, a, K_MF, **kw)
            assert int(np.sum(ref & ~cand)) == 0


def test_the_search_finds_counter_examples_when_the_margin_is_too_small():
    rng = np.random.default_rng(5)
    found = 0
    for cfg in GRAZE[:4]:
        for tol in (1e-4, 1e-6, 0.0):
            o, d, c, r2 = graze_cases(rng, 100000, *cfg, tol)
            ref, a = reference_candidate(o, d, c, r2)
            global ABS_TERM
            keep, ABS_TERM = ABS_TERM, 0.0
            try:
                cand, ok = mfma_candidate(o, d, c, r2, a, 0.25)
            finally:
                ABS_TERM = keep
            found += int(np.sum(ref & ~cand))
    assert found > 0


def test_the_filter_is_not_much_looser_than_the_one_on_the_vector_unit():
    """Candidates the exact test then rejects: within a few percent of the 7-FMA filter's (K = 256) on final.txt-like geometry."""
    from test_filter_bound import filter_candidate
    rng = np.random.default_rng(3)
    o, d, c, r2 = make_cases(rng, 400000, 10, 10, 0.05, 2, True)
    ref, a = reference_candidate(o, d, c, r2)
    mf, _ = mfma_candidate(o, d, c, r2, a, K_MF)
    vu = filter_candidate(o, d, c, r2, a, 256.0)
    assert int(np.sum(mf & ~ref)) <= 1.1 * int(np.sum(vu & ~ref)) + 100


def test_fp64_rays_through_the_fp32_operands():
    """An fp64 ray is rounded to float first (make_filter_ray_k, double overload); kFilterKMf64 covers that against the fp64 discriminant."""
    rng = np.random.default_rng(9)
    for cfg in [(10, 10, 0.05, 2, True, 1e-6), (100, 10, 0.05, 2, True, 1e-7), (10, 10, 0.05, 2, True, 0.0), (1000, 10, 0.1, 1, True, 1e-7)]:
        o64, d64, c64, r264 = make_cases64(rng, 200000, *cfg)
        ref = reference_candidate64(o64, d64, c64, r264)
        o, d = o64.astype(f32), d64.astype(f32)
        a = fma(d[:, 2], d[:, 2], fma(d[:, 1], d[:, 1], (d[:, 0] * d[:, 0]).astype(f32)))
        # (the table is packed from the fp64 records: sphere_terms takes them as they are)
        S, ok = sphere_terms(c64, r264, K_MF64)
        R = ray_terms(o, d, a, K_MF64)
        acc = np.zeros(len(o), f32)
        P = S.astype(np.float64) * R.astype(np.float64)
        for j in range(32):
            acc = (acc.astype(np.float64) + P[:, j]).astype(f32)
        cand = ~(acc < 0) | ~ok
        assert int(np.sum(ref & ~cand)) == 0


def test_margin_constants_match_the_source():
    src = open(os.path.join(ROOT, "rrt_amd", "csrc", "rrtx_device.h")).read()
    m = re.search(r"kFilterKMf = (\d+), kFilterKMf64 = (\d+)", src)
    assert m and float(m.group(1)) == K_MF and float(m.group(2)) == K_MF64
    pack = open(os.path.join(ROOT, "rrt_amd", "csrc", "rrtx_pack.h")).read()
    assert "- 1e-5L" in pack and "<= 60000.0" in pack and ">= 1e-3" in pack


def test_host_packing_matches_the_model(tmp_path):
    exe = tmp_path / "mf_pack_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "mf_pack_check.cpp"), "-o", str(exe)], check=True)
    rng = np.random.default_rng(17)
    n = 203
    c = (rng.standard_normal((n, 3)) * np.exp(rng.uniform(-3, 3.5, (n, 1)))).astype(f32)
    r = np.exp(rng.uniform(np.log(0.05), np.log(30), n)).astype(f32)
    c[0], r[0] = (0, -1000, 0), 1000  # final.txt's ground: listed apart
    c[1], r[1] = (4, 1, 0), 1
    c[2], r[2] = (300, 1, 2), 1       # a monomial beyond the f16 range
    c[3], r[3] = (1, 2, 3), 0.01      # r^2 below what the pieces resolve
    c[4], r[4] = (1, 1, 1), 250       # the threshold beyond the range
    r2 = (r * r).astype(f32)
    text = "%d\n" % n + "".join("%08x %08x %08x %08x\n" % tuple(int(x) for x in np.array([c[i, 0], c[i, 1], c[i, 2], r2[i]], f32).view(np.uint32)) for i in range(n))
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    n_pad, n_big, ok = (int(x) for x in out[0].split())
    T, in_table = sphere_terms(c, r2, K_MF)
    assert n_pad == 224 and n_big == int(np.sum(~in_table)) and ok == int(n_big <= 16) and 4 <= n_big <= 16
    assert not in_table[0] and in_table[1] and not in_table[2] and not in_table[3] and not in_table[4]
    for i in range(n_pad):
        f = out[1 + i].split()
        got = np.array([int(x, 16) for x in f[1:]], np.uint16)
        if i < n:
            assert int(f[0]) == int(not in_table[i])
            want = T[i].view(np.uint16)
        else:  # padding: never a candidate, not listed
            assert int(f[0]) == 0
            want = np.zeros(32, f16)
            want[29] = f16(60000.0)
            want = want.view(np.uint16)
        assert np.array_equal(got, want), (i, got, want)
    assert out[1 + n_pad].strip() == "roundtrip_bad 0"


def test_the_column_of_a_lane_without_a_ray_is_a_candidate_for_nothing():
    """render_kernel: a lane that does not scan hands in g = -60000 twice and thr x -1.  Against every kind of row - a sphere of the
    table (|thr| <= 60000), one listed apart, a padding record (both: thr = 60000 alone) - the sum is negative, in any order."""
    rng = np.random.default_rng(1)
    n = 20000
    c = (rng.standard_normal((n, 3)) * np.exp(rng.uniform(-3, 5.5, (n, 1)))).astype(f32)
    r = np.exp(rng.uniform(np.log(0.01), np.log(300), n)).astype(f32)
    c[:4] = [(0, 0, 0), (244, 0, 0), (0, -1000, 0), (100, 100, 100)]
    r[:4] = [244.9, 0.04, 1000, 0.5]
    S, ok = sphere_terms(c, (r * r).astype(f32), K_MF)
    assert ok.sum() > 1000 and (~ok).sum() > 1000
    col = np.zeros(32, f16)
    col[27] = col[28] = f16(-60000.0)
    col[29] = col[30] = f16(-1.0)
    P = S.astype(np.float64) * col.astype(np.float64)[None, :]
    for order in (np.arange(32), np.arange(31, -1, -1), rng.permutation(32)):
        acc = np.zeros(n, f32)
        for j in order:
            acc = (acc.astype(np.float64) + P[:, j]).astype(f32)
        assert np.all(acc <= -59000.0)
    # ... and a padding record / a sphere listed apart against any real ray: thr = 60000 times a negative power of two
    assert np.all(S[~ok][:, :29] == 0) and np.all(S[~ok][:, 29] == f16(60000.0)) and np.all(S[~ok][:, 30:] == 0)
