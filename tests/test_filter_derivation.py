"""The margins of the three conservative filters, DERIVED and held against measurement.

The test no filter may contradict is the reference's own, sphere.h:41: `discriminant < 0` -> miss, evaluated in the
precision of the build.  A filter computes, in its own arithmetic, something close to D / |d|^2 (D the discriminant in exact
arithmetic on the same operands) plus a margin  K u S,  u = 2^-24, S = |o|^2 + |c|^2 + r^2,  and says "miss" only if that
is negative.  It is safe iff       K u S  >=  |E_ref| / |d|^2  +  |E_f|,
E_ref the rounding error of the reference's discriminant, E_f that of the filter's value.  DESIGN.md 3a bounds both term
by term (standard model, first order, every constant rounded up); the budget, in units of u S:

    reference discriminant, fp32 (21 |o-c|^2 + 6 r^2 per |d|^2, and |o-c|^2 <= 2 (|o|^2 + |c|^2))      42   (fp64 rays: 0)
    fp64 rays, centres rounded to float                                                                  16   (fp32 rays: 0)
    the ray's side in fp32: n = d / |d|, s = o.n, b = 2 (o - s n), g = s^2 - |o|^2, n_i n_j              33
    vector form: c.n, the three FMAs over b.c + g, the last FMA                                          11
    matrix form: f16 x 2 operands, the low x low products dropped                                        36
    matrix form: 31 terms added in f32 in any order                                                      93
    matrix form: f16 underflow of the ray's pieces (2 u S + 2e-7 absolute)                                2
    matrix form: f16 underflow of a sphere's pieces (pack_mf_table lists a sphere apart unless
                 2^-25 U <= 9.8e-6 + 150 u (|c|^2 + r^2)  and  U <= 2.4e6,  U = the sum of its operands)  150

    vector form:  fp32 rays 42 + 33 + 11 =  86 <= kFilterK   = 256;   fp64 rays 16 + 33 + 11 =  60 <= kFilterK64   = 512
    matrix form:  fp32 rays 42 + 33 + 36 + 93 + 2 + 150 = 356 <= kFilterKMf = 512;   fp64 rays 330 <= kFilterKMf64 = 1024

This file (1) asserts the sums against the constants in the source, (2) measures every line of the budget on hostile
populations - random and grazing pairs, cameras 10 .. 30 000 units out, radii 0.04 .. 1000 - and asserts that the measured
error stays under ITS line (a line that measurement exceeds would be a wrong derivation), (3) checks the rule by which a
sphere is listed apart over the admitted operand range, and that rrtx_pack.h applies it.  That the searches find false
negatives when the margin is far too small is shown by test_filter_bound.py / test_filter_mfma.py (K <= 1)."""
import os
import re

import numpy as np
import pytest

from test_filter_bound import EPS, f32, fma, make_cases
from test_filter_mfma import graze_cases, ray_terms, sphere_terms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = EPS  # 2^-24
B_REF, B_IN64, B_RAY, B_VEC, B_SPLIT, B_SUM, B_UFLOW_RAY, B_UFLOW_SPHERE = 42, 16, 33, 11, 36, 93, 2, 150
ABS_RAY, ABS_SPHERE, ABS_TERM = 2e-7, 9.8e-6, 1e-5


def _constants():
    src = open(os.path.join(ROOT, "rrt_amd", "csrc", "rrtx_device.h")).read()
    k = int(re.search(r"constexpr int kFilterK = (\d+);", src).group(1))
    k64 = int(re.search(r"constexpr int kFilterK64 = (\d+);", src).group(1))
    m = re.search(r"kFilterKMf = (\d+), kFilterKMf64 = (\d+)", src)
    return k, k64, int(m.group(1)), int(m.group(2))


def test_the_shipped_margins_cover_the_derived_budgets():
    k, k64, kmf, kmf64 = _constants()
    assert B_REF + B_RAY + B_VEC <= k
    assert B_IN64 + B_RAY + B_VEC <= k64
    assert B_REF + B_RAY + B_SPLIT + B_SUM + B_UFLOW_RAY + B_UFLOW_SPHERE <= kmf
    assert B_IN64 + B_RAY + B_SPLIT + B_SUM + B_UFLOW_RAY + B_UFLOW_SPHERE <= kmf64
    assert ABS_RAY + ABS_SPHERE <= ABS_TERM
    # the same numbers stand in the source, as static_asserts
    pack = open(os.path.join(ROOT, "rrt_amd", "csrc", "rrtx_pack.h")).read()
    for name, v in (("kBudgetRef", B_REF), ("kBudgetIn64", B_IN64), ("kBudgetRay", B_RAY), ("kBudgetVec", B_VEC), ("kBudgetMfSplit", B_SPLIT), ("kBudgetMfSum", B_SUM),
                    ("kBudgetMfUnderflowRay", B_UFLOW_RAY), ("kBudgetMfUnderflowSphere", B_UFLOW_SPHERE)):
        assert re.search(r"constexpr int %s = %d;" % (name, v), pack), name


def _populations():
    rng = np.random.default_rng(77)
    for cfg in [(10, 10, 0.05, 2, False), (10, 10, 0.05, 2, True), (3, 3, 0.04, 30, True), (100, 10, 0.05, 2, True), (1000, 10, 0.1, 1, True), (30, 100, 0.2, 5, True)]:
        yield make_cases(rng, 60000, *cfg)
    for cfg in [(10, 10, 0.05, 2), (1000, 10, 0.1, 1), (20000, 50, 0.1, 10), (10, 1000, 900, 1100)]:
        for tol in (1e-3, 1e-6, 0.0):
            yield graze_cases(rng, 40000, *cfg, tol)


def _exact(o, d, c, r2):
    """D / |d|^2 and the magnitudes, in float64 on the float32 operands (2^-29 of the float32 roundings measured against it)."""
    o, d, c, r2 = (x.astype(np.float64) for x in (o, d, c, r2))
    w = o - c
    dd = (d * d).sum(1)
    D = (w * d).sum(1) ** 2 - dd * ((w * w).sum(1) - r2)
    return D, dd, (w * w).sum(1), (o * o).sum(1), (c * c).sum(1), r2


def _reference_disc(o, d, c, r2):
    oc = (o - c).astype(f32)
    a = ((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32)
    a = (a + (d[:, 2] * d[:, 2]).astype(f32)).astype(f32)
    hb = ((oc[:, 0] * d[:, 0]).astype(f32) + (oc[:, 1] * d[:, 1]).astype(f32)).astype(f32)
    hb = (hb + (oc[:, 2] * d[:, 2]).astype(f32)).astype(f32)
    q = ((oc[:, 0] * oc[:, 0]).astype(f32) + (oc[:, 1] * oc[:, 1]).astype(f32)).astype(f32)
    q = (q + (oc[:, 2] * oc[:, 2]).astype(f32)).astype(f32)
    return ((hb * hb).astype(f32) - (a * (q - r2).astype(f32)).astype(f32)).astype(f32), a


def test_reference_discriminant_error_is_under_its_line():
    worst = 0.0
    for o, d, c, r2 in _populations():
        disc, _ = _reference_disc(o, d, c, r2)
        D, dd, w2, o2, c2, r264 = _exact(o, d, c, r2)
        err = np.abs(disc.astype(np.float64) - D)
        bound = U * dd * (21 * w2 + 6 * r264)  # the derivation's form ...
        assert np.all(err <= bound)
        assert np.all(bound / dd <= B_REF * U * (o2 + c2 + r264) * (1 + 1e-9))  # ... and its coarsening to 42 u S
        worst = max(worst, float((err / bound).max()))
    assert 0.02 < worst <= 1.0  # (the line is a worst case, not a fit - but not vacuous either)


def _ray_side(o, d, a, K):
    inv = (f32(1) / np.sqrt(a).astype(f32)).astype(f32)
    n = (d * inv[:, None]).astype(f32)
    s = fma(o[:, 2], n[:, 2], fma(o[:, 1], n[:, 1], (o[:, 0] * n[:, 0]).astype(f32)))
    b = (f32(2) * fma(-s[:, None].repeat(3, 1), n, o)).astype(f32)
    o2 = fma(o[:, 2], o[:, 2], fma(o[:, 1], o[:, 1], (o[:, 0] * o[:, 0]).astype(f32)))
    g = fma(np.full_like(o2, f32(K * EPS)), o2, fma(s, s, -o2))
    return n, b, g


def test_ray_side_and_vector_form_errors_are_under_their_lines():
    K = 256.0
    worst_ray = worst_all = 0.0
    for o, d, c, r2 in _populations():
        _, a = _reference_disc(o, d, c, r2)
        n, b, g = _ray_side(o, d, a, K)
        D, dd, w2, o2, c2, r264 = _exact(o, d, c, r2)
        S = o2 + c2 + r264
        ideal = D / dd + (c2 - r264) + K * U * o2  # what  (c.n)^2 + b.c + g  stands for
        # the ray's side alone: the same expression in float64 from the float32 n, b, g
        n64, b64, c64 = n.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
        v_ray = (c64 * n64).sum(1) ** 2 + (b64 * c64).sum(1) + g.astype(np.float64)
        e_ray = np.abs(v_ray - ideal)
        assert np.all(e_ray <= B_RAY * U * S)
        # ... and the whole vector form as the kernel evaluates it
        u = fma(c[:, 2], n[:, 2], fma(c[:, 1], n[:, 1], (c[:, 0] * n[:, 0]).astype(f32)))
        w = fma(b[:, 2], c[:, 2], fma(b[:, 1], c[:, 1], fma(b[:, 0], c[:, 0], g)))
        e_all = np.abs(fma(u, u, w).astype(np.float64) - ideal)
        assert np.all(e_all <= (B_RAY + B_VEC) * U * S)
        worst_ray, worst_all = max(worst_ray, float((e_ray / (B_RAY * U * S)).max())), max(worst_all, float((e_all / ((B_RAY + B_VEC) * U * S)).max()))
    assert 0.02 < worst_ray <= 1.0 and 0.02 < worst_all <= 1.0


def test_fp64_operands_rounded_to_float_are_under_their_line():
    rng = np.random.default_rng(78)
    worst = 0.0
    for scale_o, scale_c, rmin, rmax in [(10, 10, 0.05, 2), (1000, 10, 0.1, 1), (30, 100, 0.2, 5), (20000, 50, 0.1, 10)]:
        n = 100000
        c = rng.standard_normal((n, 3)) * scale_c
        r2 = np.exp(rng.uniform(np.log(rmin), np.log(rmax), n)) ** 2
        o = rng.standard_normal((n, 3)) * scale_o
        d = rng.standard_normal((n, 3)) * np.exp(rng.uniform(-3, 3, (n, 1)))

        def ratio(o_, d_, c_):
            w = o_ - c_
            dd = (d_ * d_).sum(1)
            return (w * d_).sum(1) ** 2 / dd - (w * w).sum(1) + r2

        S = (o * o).sum(1) + (c * c).sum(1) + r2
        err = np.abs(ratio(o.astype(f32).astype(np.float64), d.astype(f32).astype(np.float64), c.astype(f32).astype(np.float64)) - ratio(o, d, c))
        assert np.all(err <= B_IN64 * U * S)
        worst = max(worst, float((err / (B_IN64 * U * S)).max()))
    assert 0.01 < worst <= 1.0


def _sphere_operand_sum(c64, r264, K):
    c2 = (c64 * c64).sum(1)
    v = np.abs(np.stack([c64[:, 0] ** 2, c64[:, 1] ** 2, c64[:, 2] ** 2, 2 * c64[:, 0] * c64[:, 1], 2 * c64[:, 0] * c64[:, 2], 2 * c64[:, 1] * c64[:, 2], c64[:, 0], c64[:, 1], c64[:, 2]], 1)).sum(1)
    return v + np.abs((c2 - r264) - K * U * (c2 + r264) - ABS_TERM)


def test_matrix_form_errors_are_under_their_lines():
    K = 512.0
    rng = np.random.default_rng(79)
    worst = 0.0
    for o, d, c, r2 in _populations():
        _, a = _reference_disc(o, d, c, r2)
        Sph, ok = sphere_terms(c, r2, K)
        if not ok.any():  # (the population of r = 1000 spheres: all listed apart, nothing of the table to measure)
            continue
        Ray = ray_terms(o, d, a, K)
        D, dd, w2, o2, c2, r264 = _exact(o, d, c, r2)
        S = o2 + c2 + r264
        n, b, g = _ray_side(o, d, a, K)
        m = np.maximum(np.abs(g), np.abs(b).max(1)).astype(np.float64)
        _, e = np.frexp(m)
        lam = np.ldexp(1.0, np.where(e > 14, 14 - e, 0))
        ideal = D / dd + K * U * S + ABS_TERM  # what the 31 terms stand for, divided by the scale
        P = Sph.astype(np.float64) * Ray.astype(np.float64)
        Usum = _sphere_operand_sum(c.astype(np.float64), r264, K)
        uflow = 2.0 ** -25 * Usum * np.maximum(1.0, 2 * m / 2.0 ** 14)  # the sphere's pieces' underflow, as the derivation prices it
        line = (B_RAY + B_SPLIT + B_SUM + B_UFLOW_RAY) * U * S + ABS_RAY + uflow
        for order in (np.arange(32), np.arange(31, -1, -1), rng.permutation(32)):
            acc = np.zeros(len(o), f32)
            for j in order:
                acc = (acc.astype(np.float64) + P[:, j]).astype(f32)
            err = np.abs(acc.astype(np.float64) / lam - ideal)[ok]
            assert np.all(err <= line[ok])
            worst = max(worst, float((err / line[ok]).max()))
        # (the split and the dropped products alone: the exact sum of the kept products against the float64 value of the same operands)
        n64, b64, c64 = n.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
        thr = (c2 - r264) - K * U * (c2 + r264) - ABS_TERM
        full = ((c64 * n64).sum(1) ** 2 + (b64 * c64).sum(1) + g.astype(np.float64) - thr)
        nn = np.stack([n[:, 0] * n[:, 0], n[:, 1] * n[:, 1], n[:, 2] * n[:, 2], n[:, 0] * n[:, 1], n[:, 0] * n[:, 2], n[:, 1] * n[:, 2]], 1).astype(f32).astype(np.float64)  # (the kernel's rounded n_i n_j)
        mono = np.stack([c64[:, 0] ** 2, c64[:, 1] ** 2, c64[:, 2] ** 2, 2 * c64[:, 0] * c64[:, 1], 2 * c64[:, 0] * c64[:, 2], 2 * c64[:, 1] * c64[:, 2]], 1)
        full_kernel_operands = (mono * nn).sum(1) + (b64 * c64).sum(1) + g.astype(np.float64) - thr
        e_split = np.abs(P.sum(1) / lam - full_kernel_operands)[ok]
        assert np.all(e_split <= (B_SPLIT * U * S + uflow + ABS_RAY + B_UFLOW_RAY * U * S)[ok])
        assert np.all(np.abs(full_kernel_operands - full) <= U * c2 * (1 + 1e-6))  # the rounded n_i n_j: one u of |c|^2, inside the ray's line
    assert 0.01 < worst <= 1.0


def _rule(Usum, c2, r2):
    """pack_mf_table: may the sphere stay in the table?  (the f16 underflow of ITS pieces must fit the share of the margin set aside for it)"""
    return (2.0 ** -25 * Usum <= ABS_SPHERE + B_UFLOW_SPHERE * U * (c2 + r2)) & (Usum <= 2.4e6)


def test_the_rule_that_lists_a_sphere_apart():
    # over the admitted operand range (every operand and the threshold within 60 000, r^2 >= 1e-3) the rule holds - the constants of
    # round 3 were sufficient; the rule is what says so
    rng = np.random.default_rng(80)
    n = 400000
    c = rng.standard_normal((n, 3)) * np.exp(rng.uniform(np.log(1e-3), np.log(150.0), (n, 1)))
    r2 = np.exp(rng.uniform(np.log(1e-3), np.log(6e4), n))
    c2 = (c * c).sum(1)
    Usum = _sphere_operand_sum(c, r2, 1024.0)
    thr = (c2 - r2) - 1024.0 * U * (c2 + r2) - ABS_TERM
    admitted = (np.abs(np.stack([c[:, 0] ** 2, c[:, 1] ** 2, c[:, 2] ** 2, 2 * c[:, 0] * c[:, 1], 2 * c[:, 0] * c[:, 2], 2 * c[:, 1] * c[:, 2]], 1)).max(1) <= 6e4) & (np.abs(thr) <= 6e4)
    assert admitted.sum() > 1000 and np.all(_rule(Usum, c2, r2)[admitted])
    # ... and it is not vacuous: operands that passed the caps but whose sum were larger would be listed apart
    assert not _rule(np.array([3e6]), np.array([1e4]), np.array([1.0]))[0]
    assert not _rule(np.array([1e3]), np.array([1e-6]), np.array([1e-3]))[0]
    pack = open(os.path.join(ROOT, "rrt_amd", "csrc", "rrtx_pack.h")).read()
    assert "mf_sphere_stays_in_table" in pack and "2.4e6" in pack and "9.8e-6" in pack
