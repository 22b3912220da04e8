"""The bound that lets triangles into the acceleration grid, checked numerically on the CPU (no GPU needed).

rrtx_grid.h admits a triangle to the grid if every hit the reference's Moeller-Trumbore test
(triangle.h:38-75) can report lies within  R = rho / (1 - rho) * (3 |s| + t |d| + |e1| + |e2|),
rho = 16 eps |d| |e1| |e2| / 1e-7,  of the triangle (s = o - v0).  The derivation is in DESIGN.md 3b; here
the test is restated in numpy float32 exactly as rrtx_path.h evaluates it (unfused, the reference's
operation order), run on millions of random and deliberately grazing configurations, and the residual of
every accepted hit, evaluated in float64, is compared with the bound for eps = 2^-24.  The structure of
the argument does not depend on the precision: what holds here for float32 holds for the fp64 build with
eps = 2^-53, which is where the library uses it (in float32 rho is of order one for any triangle of
practical size, so nothing is admitted - also checked here)."""
import numpy as np

f32 = np.float32
EPS32 = 2.0 ** -24
CUT = f32(0.0000001)


def cross(a, b):
    return np.stack([(a[:, 1] * b[:, 2]).astype(f32) - (a[:, 2] * b[:, 1]).astype(f32), (a[:, 2] * b[:, 0]).astype(f32) - (a[:, 0] * b[:, 2]).astype(f32),
                     (a[:, 0] * b[:, 1]).astype(f32) - (a[:, 1] * b[:, 0]).astype(f32)], axis=1).astype(f32)


def dot(a, b):  # vec3.h:101-104: (x*x' + y*y') + z*z'
    return (((a[:, 0] * b[:, 0]).astype(f32) + (a[:, 1] * b[:, 1]).astype(f32)).astype(f32) + (a[:, 2] * b[:, 2]).astype(f32)).astype(f32)


def moeller_trumbore(o, d, v0, e1, e2):
    """triangle_test<float, true> of rrtx_path.h (= triangle.h:38-75) without the t_min / t_max window."""
    h = cross(d, e2)
    a = dot(e1, h)
    ok = ~((a > -CUT) & (a < CUT))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        f = (f32(1.0) / a).astype(f32)
        s = (o - v0).astype(f32)
        u = dot((f[:, None] * s).astype(f32), h)
        ok &= ~((u < 0) | (u > 1))
        q = cross(s, e1)
        v = dot((f[:, None] * d).astype(f32), q)
        ok &= ~((v < 0) | ((u + v).astype(f32) > 1))
        t = dot((f[:, None] * e2).astype(f32), q)
    ok &= t > CUT
    return ok, t, u, v


def cases(rng, n, grazing):
    size = np.exp(rng.uniform(np.log(1e-3), np.log(0.3), (n, 1)))
    v0 = rng.uniform(-20, 20, (n, 3))
    e1 = rng.standard_normal((n, 3))
    e1 *= size / np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = rng.standard_normal((n, 3))
    e2 *= size * rng.uniform(0.3, 1.5, (n, 1)) / np.linalg.norm(e2, axis=1, keepdims=True)
    bu, bv = rng.uniform(0, 1, (n, 1)), rng.uniform(0, 1, (n, 1))
    target = v0 + bu * e1 + bv * (1 - bu) * e2  # a point of the triangle
    dist = np.exp(rng.uniform(np.log(0.05), np.log(60.0), (n, 1)))
    if grazing:  # from a point in (nearly) the triangle's plane: |d . N| from 1e-9 to 1e-2
        w = rng.standard_normal((n, 3))
        nrm = np.cross(e1, e2)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        w -= (w * nrm).sum(1, keepdims=True) * nrm
        w /= np.linalg.norm(w, axis=1, keepdims=True)
        tilt = np.exp(rng.uniform(np.log(1e-9), np.log(1e-2), (n, 1))) * rng.choice([-1.0, 1.0], (n, 1))
        o = target - dist * (w + tilt * nrm)
    else:
        w = rng.standard_normal((n, 3))
        w /= np.linalg.norm(w, axis=1, keepdims=True)
        o = target - dist * w
    d = (target - o) * np.exp(rng.uniform(np.log(0.05), np.log(3.0), (n, 1))) / dist
    return [x.astype(f32) for x in (o, d, v0, e1, e2)]


def residual_and_bound(o, d, v0, e1, e2, t, u, v):
    o, d, v0, e1, e2 = (x.astype(np.float64) for x in (o, d, v0, e1, e2))
    t, u, v = (x.astype(np.float64)[:, None] for x in (t, u, v))
    res = np.linalg.norm(o + t * d - v0 - u * e1 - v * e2, axis=1)
    nd, n1, n2, ns = (np.linalg.norm(x, axis=1) for x in (d, e1, e2, o - v0))
    rho = 16 * EPS32 * nd * n1 * n2 / 1e-7
    with np.errstate(divide="ignore", invalid="ignore"):
        bound = rho / (1 - rho) * (3 * ns + t[:, 0] * nd + n1 + n2)
    return res, bound, rho


def test_accepted_hits_lie_within_the_residual_bound():
    rng = np.random.default_rng(17)
    worst, counted, hard = 0.0, 0, 0
    for grazing in (False, True, True, True):
        o, d, v0, e1, e2 = cases(rng, 1500000, grazing)
        ok, t, u, v = moeller_trumbore(o, d, v0, e1, e2)
        res, bound, rho = residual_and_bound(o, d, v0, e1, e2, t, u, v)
        sel = ok & (rho < 0.5) & np.isfinite(res)
        counted += int(sel.sum())
        hard += int((sel & (res > 0.01 * bound)).sum())
        assert not np.any(res[sel] > bound[sel]), float(np.max(res[sel] / bound[sel]))
        worst = max(worst, float(np.max(res[sel] / bound[sel])))
    assert counted > 500000 and worst < 1.0  # (the worst observed ratio is ~0.05: the constant 16 is generous)
    assert hard > 100  # the search reaches the regime where the residual is a visible fraction of the bound


def test_the_residual_is_real_and_float32_admits_nothing():
    """Grazing hits do land far from the triangle in float32 - more than a triangle's size away - which is why
    a box inflated by a fixed fraction of a cell cannot be conservative there; and the admission rule of
    rrtx_grid.h (rho <= 1e-4 at |d| = 1000) rejects every triangle of practical size in float32 while it
    admits them all in float64."""
    rng = np.random.default_rng(5)
    o, d, v0, e1, e2 = cases(rng, 2000000, True)
    ok, t, u, v = moeller_trumbore(o, d, v0, e1, e2)
    res, bound, rho = residual_and_bound(o, d, v0, e1, e2, t, u, v)
    size = np.linalg.norm(e1.astype(np.float64), axis=1)
    assert int((ok & (res > size)).sum()) > 10
    e1e2 = np.array([1e-3 * 1e-3, 0.03 * 0.03, 0.3 * 0.3, 2.0 * 2.0])  # |e1| |e2|
    assert not np.any(16 * EPS32 * 1000 * e1e2 / 1e-7 <= 1e-4)
    assert np.all(16 * 2.0 ** -53 * 1000 * e1e2 / 1e-7 <= 1e-4)
