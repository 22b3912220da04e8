"""ctypes front-ends for the TEST-ONLY checkers under oracle/.

* ``Oracle``    - oracle/librrt_oracle.so, the CPU restatement (oracle/rrt_oracle.cpp).
* ``Reference`` - oracle/_ref/libref_f32.so / libref_f64.so, the reference's own sources compiled
  unchanged with the product's RNG hooked in (oracle/ref_harness.cpp).  Exists only where
  oracle/_ref was built (this container; it travels to the GPU box as a prebuilt file).

Nothing in the product (rrt_amd/, include/) imports this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librrt_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def scene_path(name):
    return os.path.join(SCENES, name if name.endswith(".txt") else name + ".txt")


def build_oracle():
    """(Re)build oracle/librrt_oracle.so when missing or stale (CPU only, ~5 s)."""
    src = os.path.join(ORACLE_DIR, "rrt_oracle.cpp")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "librrt_oracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def have_reference():
    return os.path.exists(os.path.join(REF_DIR, "libref_f32.so")) and os.path.exists(os.path.join(REF_DIR, "libref_f64.so"))


def _np(fp64):
    return np.float64 if fp64 else np.float32


class _Tables:
    """cam24 / materials(n,6) / spheres(n,5) / msph(n,10) / tris(n,10) as float64 arrays."""

    def __init__(self, counts, dump):
        nm, ns, nms, nt = counts[0], counts[1], counts[2], counts[3]
        self.counts = list(counts)
        self.cam = np.zeros(24)
        self.materials = np.zeros((nm, 6))
        self.spheres = np.zeros((ns, 5))
        self.msph = np.zeros((nms, 10))
        self.tris = np.zeros((nt, 10))
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        dump(p(self.cam), p(self.materials), p(self.spheres), p(self.msph), p(self.tris))


class Oracle:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(build_oracle())
            L.rrto_scene_load.restype = C.c_void_p
            L.rrto_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
            L.rrto_scene_free.argtypes = [C.c_void_p]
            L.rrto_scene_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            L.rrto_scene_dump.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 5
            L.rrto_render.restype = C.c_int
            L.rrto_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
            L.rrto_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
            L.rrto_quantise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.rrto_convert_color.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_int)]
            L.rrto_rng_words.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)]
            L.rrto_rng_f32.restype = C.c_float
            L.rrto_rng_f32.argtypes = [C.c_uint32] * 4
            L.rrto_rng_f64.restype = C.c_double
            L.rrto_rng_f64.argtypes = [C.c_uint32] * 4
            cls._lib = L
        return cls._lib

    def __init__(self, scene_file, w, h, fp64=False):
        L = self.lib()
        err = C.c_int(0)
        self.fp64 = bool(fp64)
        self.w, self.h = w, h
        self.h_ = L.rrto_scene_load(scene_file.encode(), w, h, int(fp64), C.byref(err))
        self.err = err.value
        if not self.h_:
            raise ValueError("oracle: scene load failed with reference exit code %d" % self.err)

    def __del__(self):
        if getattr(self, "h_", None):
            self.lib().rrto_scene_free(self.h_)
            self.h_ = None

    def counts(self):
        c = (C.c_int * 6)()
        self.lib().rrto_scene_counts(self.h_, c)
        return list(c)

    def tables(self):
        return _Tables(self.counts(), lambda *a: self.lib().rrto_scene_dump(self.h_, *a))

    def render(self, spp, depth=50, seed=1984, order=1, chunk=0, rows=None):
        """-> (fb[h,w,3] row 0 = bottom, stats dict). order 0 = recursive (rrt.cpp), 1 = iterative (rrt.cu)."""
        fb = np.zeros((self.h, self.w, 3), dtype=_np(self.fp64))
        st = (C.c_uint64 * 4)()
        r0, r1 = rows if rows else (0, self.h)
        rc = self.lib().rrto_render(self.h_, self.w, self.h, spp, depth, seed, order, chunk, r0, r1, fb.ctypes.data_as(C.c_void_p), st)
        assert rc == 0
        return fb, dict(segments=st[0], prim_tests=st[1], draws=st[2], samples=st[3])

    def sample(self, i, j, s, depth=50, seed=1984, order=1):
        out = (C.c_double * 3)()
        st = (C.c_uint64 * 4)()
        self.lib().rrto_sample(self.h_, self.w, self.h, i, j, s, depth, seed, order, out, st)
        return np.array(list(out)), dict(segments=st[0], prim_tests=st[1], draws=st[2])

    @classmethod
    def quantise(cls, fb, spp):
        fp64 = fb.dtype == np.float64
        h, w, _ = fb.shape
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        fbc = np.ascontiguousarray(fb)
        cls.lib().rrto_quantise(fbc.ctypes.data_as(C.c_void_p), int(fp64), w, h, spp, rgb.ctypes.data_as(C.c_void_p))
        return rgb

    @classmethod
    def convert_color(cls, rgb, spp, fp64=False):
        a = (C.c_double * 3)(*[float(x) for x in rgb])
        o = (C.c_int * 3)()
        cls.lib().rrto_convert_color(a, int(fp64), spp, o)
        return list(o)

    @classmethod
    def rng_words(cls, seed, pixel, sample, n):
        o = (C.c_uint32 * 4)()
        cls.lib().rrto_rng_words(seed, pixel, sample, n, o)
        return list(o)  # k0, k1, hi, lo

    @classmethod
    def rng_uniform(cls, seed, pixel, sample, n, fp64=False):
        L = cls.lib()
        return L.rrto_rng_f64(seed, pixel, sample, n) if fp64 else L.rrto_rng_f32(seed, pixel, sample, n)


class Reference:
    """The reference's own code (compiled into oracle/_ref by `make -C oracle ref`)."""

    _libs = {}

    @classmethod
    def lib(cls, fp64):
        key = bool(fp64)
        if key not in cls._libs:
            L = C.CDLL(os.path.join(REF_DIR, "libref_f64.so" if fp64 else "libref_f32.so"))
            L.ref_scene_load.restype = C.c_void_p
            L.ref_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
            L.ref_scene_free.argtypes = [C.c_void_p]
            L.ref_scene_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            L.ref_sizeof.restype = C.c_int
            L.ref_sizeof.argtypes = [C.c_int]
            L.ref_scene_raw.argtypes = [C.c_void_p] * 6
            L.ref_scene_dump.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 5
            L.ref_render.restype = C.c_int
            L.ref_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.ref_quantise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.ref_convert_color.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
            L.ref_ppm.restype = C.c_int
            L.ref_ppm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
            L.ref_rng_probe.restype = C.c_double
            L.ref_rng_probe.argtypes = [C.c_uint32] * 4
            cls._libs[key] = L
        return cls._libs[key]

    def __init__(self, scene_file, w, h, fp64=False):
        self.fp64 = bool(fp64)
        self.L = self.lib(fp64)
        self.w, self.h = w, h
        self.h_ = self.L.ref_scene_load(scene_file.encode(), w, h)

    def counts(self):
        c = (C.c_int * 6)()
        self.L.ref_scene_counts(self.h_, c)
        return list(c)

    def sizeof(self):
        return [self.L.ref_sizeof(k) for k in range(6)]

    def tables(self):
        return _Tables(self.counts(), lambda *a: self.L.ref_scene_dump(self.h_, *a))

    def raw(self):
        """Raw bytes of the reference's POD tables (what rrt.cu:217-247 marshals)."""
        cnt = self.counts()
        sz = self.sizeof()
        bufs = [np.zeros(sz[1], np.uint8), np.zeros(cnt[0] * sz[2], np.uint8), np.zeros(cnt[1] * sz[3], np.uint8), np.zeros(cnt[2] * sz[4], np.uint8),
                np.zeros(cnt[3] * sz[5], np.uint8)]
        self.L.ref_scene_raw(self.h_, *[b.ctypes.data_as(C.c_void_p) if b.size else None for b in bufs])
        return bufs

    def render(self, spp, depth=50, seed=1984, bvh=False, rows=None):
        fb = np.zeros((self.h, self.w, 3), dtype=_np(self.fp64))
        r0, r1 = rows if rows else (0, self.h)
        rc = self.L.ref_render(self.h_, self.w, self.h, spp, depth, seed, int(bvh), r0, r1, fb.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return fb

    def quantise(self, fb, spp):
        h, w, _ = fb.shape
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        fbc = np.ascontiguousarray(fb, dtype=_np(self.fp64))
        self.L.ref_quantise(fbc.ctypes.data_as(C.c_void_p), w, h, spp, rgb.ctypes.data_as(C.c_void_p))
        return rgb

    def convert_color(self, rgb, spp):
        a = (C.c_double * 3)(*[float(x) for x in rgb])
        o = (C.c_int * 3)()
        self.L.ref_convert_color(a, spp, o)
        return list(o)

    def ppm(self, fb, spp, path):
        h, w, _ = fb.shape
        fbc = np.ascontiguousarray(fb, dtype=_np(self.fp64))
        assert self.L.ref_ppm(fbc.ctypes.data_as(C.c_void_p), w, h, spp, path.encode()) == 0

    def rng_probe(self, seed, pixel, sample, n):
        return self.L.ref_rng_probe(seed, pixel, sample, n)


from mesh_gen import mesh_scene  # noqa: E402,F401  (tools/mesh_gen.py: the UV-sphere mesh scene, shared with bench.py's `configs.mesh`)


def random_scene(rng, path):
    """A valid scene file exercising the whole grammar of scene.h:224-428 with random content: number
    formats std::stod accepts, optional shutter times, duplicate and unknown material names, ignored lines, objs with
    several instances and transform chains in random order."""
    def num(lo, hi):
        v = float(rng.uniform(lo, hi))
        return rng.choice(["%r" % v, "%.3f" % v, "%.6e" % v, "%+.4f" % v, "%d" % int(round(v)) if abs(v) >= 1 else "%.2f" % v])

    lines = ["# random scene", "", "  sphere 0 0 0 1 indented_lines_are_ignored"]
    cam = "camera %s %s %s  %s %s %s  0 1 0  %s %s %s" % (num(3, 14), num(1, 5), num(2, 9), num(-1, 1), num(-0.5, 1), num(-1, 1), num(15, 70), num(0, 0.3), num(4, 14))
    if rng.random() < 0.5:
        cam += " %s %s" % (num(0, 0.2), num(0.3, 1.0))
    lines.append(cam)
    names = []
    for k in range(int(rng.integers(2, 9))):
        name = "m%d" % (k if rng.random() > 0.15 or not names else 0)  # now and then a duplicate name
        kind = rng.choice(["lambertian", "metal", "dielectric"])
        if kind == "lambertian":
            lines.append("material %s lambertian %s %s %s" % (name, num(0, 1), num(0, 1), num(0, 1)))
        elif kind == "metal":
            lines.append("material %s metal %s %s %s %s" % (name, num(0, 1), num(0, 1), num(0, 1), num(0, 1.5)))  # fuzz > 1 is clamped later
        else:
            lines.append("material %s dielectric %s" % (name, num(1.1, 2.4)))
        names.append(name)
        if rng.random() < 0.3:
            lines.append("#material not_a_material lambertian 1 1 1")
    pick = lambda: rng.choice(names) if rng.random() > 0.1 else "no_such_material"  # (maps to index 0, scene.h:310)
    lines.append("sphere 0 -1000 0 1000 %s" % names[0])
    for _ in range(int(rng.integers(1, 12))):
        lines.append("sphere %s %s %s %s %s" % (num(-6, 6), num(0.1, 1.5), num(-6, 6), num(0.05, 1.2), pick()))
    for _ in range(int(rng.integers(0, 4))):
        lines.append("msphere %s %s %s  %s %s %s  %s %s  %s %s" % (num(-4, 4), num(0.2, 1), num(-4, 4), num(-4, 4), num(0.2, 1.5), num(-4, 4), num(0, 0.3), num(0.5, 1), num(0.1, 0.5), pick()))
    n_obj = int(rng.integers(0, 3))
    for _ in range(n_obj):
        nv, nt = int(rng.integers(3, 9)), int(rng.integers(1, 7))
        lines.append("obj_beg %d %d" % (nv, nt))
        lines += ["obj_vtx %s %s %s" % (num(-1, 1), num(-1, 1), num(-1, 1)) for _ in range(nv)]
        lines += ["obj_tri %d %d %d" % tuple(int(x) for x in rng.choice(nv, 3, replace=False)) for _ in range(nt)]
        lines.append("obj_end")
    for _ in range(int(rng.integers(0, 4)) if n_obj else 0):
        inst = "obj %d %s" % (int(rng.integers(0, n_obj)), pick())
        for _ in range(int(rng.integers(0, 4))):
            kind = rng.choice(["t", "s", "r"])
            if kind == "r":
                ax = rng.normal(size=3)
                ax /= np.linalg.norm(ax)
                inst += " r %s %r %r %r" % (num(-180, 180), float(ax[0]), float(ax[1]), float(ax[2]))
            else:
                inst += " %s %s %s %s" % (kind, num(-2, 2) if kind == "t" else num(0.3, 2), num(-2, 2) if kind == "t" else num(0.3, 2), num(-2, 2) if kind == "t" else num(0.3, 2))
        lines.append(inst)
    open(path, "w").write("\n".join(str(x) for x in lines) + "\n")


def crowded_scene(rng, path, spheres_only=None):
    """Random scenes with enough primitives for the acceleration grid (40 - 600): spheres over four orders of magnitude of
    size - clusters of tiny ones, a few that dwarf a cell, some far from everything -, moving spheres, and one or two small
    meshes instanced several times under random transforms (a third of the scenes: spheres alone); a camera anywhere, sometimes
    inside the crowd, sometimes far away."""
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    lines = []
    far_cam = rng.random() < 0.2
    draw = rng.random() < 0.35
    spheres_only = draw if spheres_only is None else spheres_only  # (neither moving spheres nor meshes: the kernels' variants for such scenes)
    span = u(2, 12)  # the crowd lives in [-span, span]^2 x [0, span / 2]
    cd = u(40, 400) if far_cam else u(0.5, 2.5) * span
    ang, el = u(0, 2 * np.pi), u(0.05, 1.2)
    cam = (float(cd * np.cos(ang) * np.cos(el)), float(cd * np.sin(el) + 0.2), float(cd * np.sin(ang) * np.cos(el)))
    lines.append("camera %r %r %r  %r %r %r  0 1 0  %r %r %r%s" % (cam[0], cam[1], cam[2], u(-1, 1), u(0, 1), u(-1, 1), u(3, 12) if far_cam else u(20, 75), u(0, 0.2), max(0.5, cd * u(0.6, 1.2)),
                                                               " 0.0 1.0" if rng.random() < 0.5 else ""))
    lines += ["material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 %r" % u(0, 0.6), "material g dielectric 1.5", "material r lambertian 0.8 0.2 0.2", "material k metal 0.9 0.7 0.3 0.0"]
    mats = "amgrk"
    pick = lambda: mats[int(rng.integers(len(mats)))]
    if rng.random() < 0.85:
        lines.append("sphere 0 -1000 0 1000 a")
    n = int(rng.choice([40, 60, 120, 300, 600]))
    base = 10 ** u(-1.6, -0.3) * span / 4  # typical radius
    for k in range(n):
        kind = rng.random()
        r = base * 10 ** u(-0.3, 0.3)
        if kind < 0.03:
            r = base * 10 ** u(0.8, 1.6)  # dwarfs a cell
        elif kind < 0.08:
            r = base * 10 ** u(-2.5, -1.2)  # tiny
        x, z = u(-span, span), u(-span, span)
        y = r if rng.random() < 0.6 else u(0, span / 2)
        if kind > 0.97:
            x, z = x * 8, z * 8  # far from everything
        if rng.random() < 0.06 and not spheres_only:
            lines.append("msphere %r %r %r  %r %r %r  0.0 1.0  %r %s" % (x, y, z, x + u(-1, 1) * r * 3, y + u(0, 1) * r * 3, z + u(-1, 1) * r * 3, r, pick()))
        else:
            lines.append("sphere %r %r %r %r %s" % (x, y, z, r if rng.random() > 0.02 else -r, pick()))
    if rng.random() < 0.3:  # a cluster of coincident / nested spheres
        cx, cz, cr = u(-span, span), u(-span, span), base
        lines += ["sphere %r %r %r %r g" % (cx, cr, cz, cr), "sphere %r %r %r %r g" % (cx, cr, cz, -0.9 * cr), "sphere %r %r %r %r m" % (cx, cr, cz, cr), "sphere %r %r %r %r r" % (cx, cr, cz, 0.5 * cr)]
    n_obj = 0 if spheres_only else int(rng.integers(0, 3))
    for _ in range(n_obj):
        nu, nv = int(rng.integers(2, 7)), int(rng.integers(3, 12))
        verts = [(np.sin(np.pi * i / nu) * np.cos(2 * np.pi * j / nv), np.cos(np.pi * i / nu), np.sin(np.pi * i / nu) * np.sin(2 * np.pi * j / nv)) for i in range(nu + 1) for j in range(nv)]
        tris = []
        for i in range(nu):
            for j in range(nv):
                a, b, c, d = i * nv + j, i * nv + (j + 1) % nv, (i + 1) * nv + j, (i + 1) * nv + (j + 1) % nv
                if i > 0:
                    tris.append((a, c, b))
                if i < nu - 1:
                    tris.append((b, c, d))
        lines.append("obj_beg %d %d" % (len(verts), len(tris)))
        lines += ["obj_vtx %r %r %r" % tuple(float(x) for x in v) for v in verts]
        lines += ["obj_tri %d %d %d" % t for t in tris]
        lines.append("obj_end")
    for _ in range(int(rng.integers(1, 6)) if n_obj else 0):
        sc = base * 10 ** u(0.2, 1.2)
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        lines.append("obj %d %s s %r %r %r r %r %r %r %r t %r %r %r" % (int(rng.integers(n_obj)), pick(), sc * u(0.5, 1.5), sc * u(0.5, 1.5), sc * u(0.5, 1.5), u(-180, 180), float(ax[0]), float(ax[1]), float(ax[2]),
                                                                      u(-span, span), sc + u(0, span / 3), u(-span, span)))
    open(str(path), "w").write("\n".join(lines) + "\n")
    return str(path)


def degenerate_scene(path):
    """What a scene file can legally contain and a renderer trips over: a sphere of radius 0, the hollow-glass idiom of a
    NEGATIVE radius (sphere.h:38,53: hit by r*r, normal (p - c) / r flipped), two identical spheres (every hit a tie), a
    triangle of zero area (collinear vertices: its unit normal is 0/0 - never hit, |a| < 1e-7, but packed and gridded), a
    triangle with a repeated vertex, a moving sphere that does not move and one whose own time interval is empty, in a camera
    with a shutter."""
    lines = ["camera 0 1.5 7 0 0.8 0 0 1 0 35 0.08 7 0.0 1.0", "material a lambertian 0.7 0.4 0.3", "material g dielectric 1.5", "material m metal 0.8 0.8 0.9 0.0",
             "sphere 0 -100 0 100 a", "sphere 0 1 0 1.0 g", "sphere 0 1 0 -0.9 g", "sphere -2.2 0.7 0.5 0.7 m", "sphere -2.2 0.7 0.5 0.7 a", "sphere 2 0.5 1 0 a",
             "msphere 2.2 0.6 0.3  2.2 0.6 0.3  0 1  0.6 m", "msphere -0.8 0.3 2.5  -0.2 0.5 2.5  0.5 0.5  0.3 a",
             "obj_beg 6 3", "obj_vtx 0 0 0", "obj_vtx 1 0 0", "obj_vtx 2 0 0", "obj_vtx 0 1 0", "obj_vtx 1 1 0.2", "obj_vtx 0.3 0.2 0.9",
             "obj_tri 0 1 2", "obj_tri 0 0 3", "obj_tri 3 4 5", "obj_end", "obj 0 a t 0.5 0.3 3", "obj 0 m s 0.8 0.8 0.8 r 30 0 1 0 t -2 0.2 3"]
    open(str(path), "w").write("\n".join(lines) + "\n")
    return str(path)
