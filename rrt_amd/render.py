"""Scene / Rrt: the reference's two host-visible classes (scene.h:210, rrt.h:14) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

_TABLE_DTYPES = {}


def _table_dtypes(fp64):
    """numpy views of the reference-layout POD tables (include/rrtx.h)."""
    key = bool(fp64)
    if key not in _TABLE_DTYPES:
        f = np.float64 if fp64 else np.float32
        fs = 8 if fp64 else 4
        camera = np.dtype([("v", f, (7, 3)), ("lens_radius", f), ("time0", f), ("time1", f)])
        fuzz_off = 8 + 3 * fs
        fuzz_off = (fuzz_off + 7) // 8 * 8
        material = np.dtype({"names": ["type", "albedo", "fuzz", "ref_idx"], "formats": [np.int32, (f, 3), np.float64, np.float64], "offsets": [0, 8, fuzz_off, 8], "itemsize": fuzz_off + 8})
        r_off = (3 * fs + 7) // 8 * 8
        sphere = np.dtype({"names": ["center", "radius", "material_idx"], "formats": [(f, 3), np.float64, np.int32], "offsets": [0, r_off, r_off + 8], "itemsize": r_off + 16})
        t_off = (6 * fs + 7) // 8 * 8
        msphere = np.dtype({"names": ["center0", "center1", "time0", "time1", "radius", "material_idx"], "formats": [(f, 3), (f, 3), np.float64, np.float64, np.float64, np.int32],
                            "offsets": [0, 3 * fs, t_off, t_off + 8, t_off + 16, t_off + 24], "itemsize": t_off + 32})
        tri_size = (9 * fs + 4 + fs - 1) // fs * fs
        triangle = np.dtype({"names": ["vertices", "material_idx"], "formats": [(f, (3, 3)), np.int32], "offsets": [0, 9 * fs], "itemsize": tri_size})
        _TABLE_DTYPES[key] = dict(camera=camera, material=material, sphere=sphere, msphere=msphere, triangle=triangle)
    return _TABLE_DTYPES[key]


class Scene:
    """scene(filename, image_width, image_height) - scene.h:212.  Parsed by librrtx's host code."""

    def __init__(self, filename, image_width, image_height, fp64=False):
        self.fp64 = bool(fp64)
        self.image_width, self.image_height = int(image_width), int(image_height)
        h = C.c_void_p()
        rc = lib.rrtx_scene_load(str(filename).encode(), self.image_width, self.image_height, int(self.fp64), C.byref(h))
        if rc != 0:
            # the reference exits the process with this code (scene.h:222,289,433-441)
            self.exit_code = lib.rrtx_scene_exit_code()
            raise ValueError("scene %r rejected (reference exit code %d)" % (filename, self.exit_code))
        self._h = h
        self.desc = _lib.SceneDesc()
        check(lib.rrtx_scene_describe(self._h, C.byref(self.desc)), "rrtx_scene_describe")

    @classmethod
    def from_tables(cls, camera, materials, spheres=None, moving_spheres=None, triangles=None, fp64=False):
        """Build a scene straight from reference-layout tables (numpy structured arrays)."""
        self = cls.__new__(cls)
        self.fp64 = bool(fp64)
        self._h = None
        dt = _table_dtypes(fp64)
        self._keep = [np.ascontiguousarray(camera, dtype=dt["camera"]).reshape(1), np.ascontiguousarray(materials, dtype=dt["material"]),
                      np.ascontiguousarray(spheres if spheres is not None else [], dtype=dt["sphere"]),
                      np.ascontiguousarray(moving_spheres if moving_spheres is not None else [], dtype=dt["msphere"]),
                      np.ascontiguousarray(triangles if triangles is not None else [], dtype=dt["triangle"])]
        d = _lib.SceneDesc()
        d.fp64 = int(self.fp64)
        ptr = lambda a: a.ctypes.data if a.size else None
        d.camera = ptr(self._keep[0])
        d.num_materials, d.materials = len(self._keep[1]), ptr(self._keep[1])
        d.num_spheres, d.spheres = len(self._keep[2]), ptr(self._keep[2])
        d.num_moving_spheres, d.moving_spheres = len(self._keep[3]), ptr(self._keep[3])
        d.num_triangles, d.triangles = len(self._keep[4]), ptr(self._keep[4])
        self.desc = d
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.rrtx_scene_free(h)
            self._h = None

    def counts(self):
        """materials, spheres, moving spheres, instance triangles, objs, obj instances."""
        if self._h is None:
            d = self.desc
            return [d.num_materials, d.num_spheres, d.num_moving_spheres, d.num_triangles, 0, 0]
        c = (C.c_int32 * 6)()
        check(lib.rrtx_scene_counts(self._h, c), "rrtx_scene_counts")
        return list(c)

    def tables(self):
        """Copies of the POD tables as numpy structured arrays (reference layouts)."""
        dt = _table_dtypes(self.fp64)
        d = self.desc

        def view(ptr, n, dtype):
            if not ptr or n == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=n).copy()

        return dict(camera=view(d.camera, 1, dt["camera"])[0], materials=view(d.materials, d.num_materials, dt["material"]), spheres=view(d.spheres, d.num_spheres, dt["sphere"]),
                    moving_spheres=view(d.moving_spheres, d.num_moving_spheres, dt["msphere"]), triangles=view(d.triangles, d.num_triangles, dt["triangle"]))


class Rrt:
    """class Rrt (rrt.h:14-48): Rrt(image_width, image_height, samples_per_pixel, max_depth, use_bvh,
    threads_x, threads_y) and render(scene) -> framebuffer (row 0 = bottom, un-normalised sums)."""

    def __init__(self, image_width, image_height, samples_per_pixel, max_depth, use_bvh=True, threads_x=8, threads_y=8, *, fp64=False, device=0, seed=1984,
                 sample_chunk=0, shard_rank=0, shard_count=1, tile_rows=4, collect_stats=True, exact_scan=False, flags=0, handoff_lanes=0, handoff_iters=0, list_passes=0, taper_samples=0):
        p = _lib.Params()
        p.image_width, p.image_height = int(image_width), int(image_height)
        p.samples_per_pixel, p.max_depth = int(samples_per_pixel), int(max_depth)
        p.use_bvh = int(bool(use_bvh))
        p.threads_x, p.threads_y = int(threads_x), int(threads_y)
        p.fp64 = int(bool(fp64))
        p.device = int(device)
        p.seed = int(seed)
        p.sample_chunk = int(sample_chunk)
        p.shard_rank, p.shard_count, p.tile_rows = int(shard_rank), int(shard_count), int(tile_rows)
        p.collect_stats = int(bool(collect_stats))
        p.flags = (_lib.FLAG_EXACT_SCAN if exact_scan else 0) | int(flags)
        p.handoff_lanes = int(handoff_lanes)
        p.handoff_iters = int(handoff_iters)
        p.list_passes = int(list_passes)
        p.taper_samples = int(taper_samples)
        self.params = p
        self.fp64 = bool(fp64)
        self._ctx = C.c_void_p()
        check(lib.rrtx_create(C.byref(p), C.byref(self._ctx)), "rrtx_create")
        self._scene = None
        self.stats = None

    def close(self):
        ctx = getattr(self, "_ctx", None)
        if ctx:
            lib.rrtx_destroy(ctx)
            self._ctx = None

    __del__ = close

    @property
    def dtype(self):
        return np.float64 if self.fp64 else np.float32

    def shard_rows(self):
        n = lib.rrtx_shard_rows(self._ctx, None, 0)
        rows = (C.c_int32 * max(n, 1))()
        lib.rrtx_shard_rows(self._ctx, rows, n)
        return np.array(rows[:n], dtype=np.int32)

    def set_scene(self, scene):
        check(lib.rrtx_set_scene(self._ctx, C.byref(scene.desc)), "rrtx_set_scene")
        self._scene = scene

    def render(self, scene=None):
        """vec3* Rrt::render(scene*) - returns fb[h, w, 3]; only this shard's rows are filled."""
        if scene is not None:
            self.set_scene(scene)
        p = self.params
        fb = np.zeros((p.image_height, p.image_width, 3), dtype=self.dtype)
        st = _lib.Stats()
        check(lib.rrtx_render(self._ctx, fb.ctypes.data_as(C.c_void_p), C.byref(st)), "rrtx_render")
        self.stats = st.as_dict()
        return fb

    def render_device(self, device_ptr, stream_ptr=0):
        """Enqueue one render into device memory (local rows, compact).  `device_ptr` / `stream_ptr`
        are raw addresses, e.g. tensor.data_ptr() and torch.cuda.current_stream().cuda_stream
        (0 = HIP's null stream, which is also torch's default stream)."""
        check(lib.rrtx_render_device(self._ctx, C.c_void_p(int(device_ptr)), C.c_void_p(int(stream_ptr))), "rrtx_render_device")

    @property
    def stream(self):
        """The context's own hipStream_t (used by render())."""
        return lib.rrtx_stream(self._ctx) or 0

    def collect(self):
        st = _lib.Stats()
        check(lib.rrtx_collect(self._ctx, C.byref(st)), "rrtx_collect")
        self.stats = st.as_dict()
        return self.stats


class RrtGroup:
    """One frame over several devices of one node, driven by this one process (include/rrtx.h, rrtx_group): row-tile
    shards, one grouped RCCL send / recv to the first device, de-interleave there.  `devices` is a count (ordinals
    0 .. n-1) or a list of ordinals; a list that repeats an ordinal needs rehearsal=True (the N-way decomposition
    rehearsed on fewer GPUs: device-to-device copies instead of RCCL).  Same image as Rrt, bit for bit."""

    def __init__(self, devices, image_width, image_height, samples_per_pixel, max_depth, use_bvh=True, *, fp64=False, seed=1984, sample_chunk=0, tile_rows=4,
                 collect_stats=True, flags=0, rehearsal=False):
        devs = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        p = _lib.Params()
        p.image_width, p.image_height = int(image_width), int(image_height)
        p.samples_per_pixel, p.max_depth = int(samples_per_pixel), int(max_depth)
        p.use_bvh = int(bool(use_bvh))
        p.threads_x = p.threads_y = 8
        p.fp64 = int(bool(fp64))
        p.seed = int(seed)
        p.sample_chunk = int(sample_chunk)
        p.tile_rows = int(tile_rows)
        p.collect_stats = int(bool(collect_stats))
        p.flags = int(flags)
        self.params, self.fp64, self.devices = p, bool(fp64), devs
        self._g = C.c_void_p()
        arr = (C.c_int32 * len(devs))(*devs)
        # RCCL greets on the process's stdout while the communicators are built; a host that owns stdout (bench.py prints its one
        # JSON line there) keeps it clean itself - the library does not touch file descriptors (include/rrtx.h, ABI 4)
        import os
        import sys

        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            rc = lib.rrtx_group_create(C.byref(p), len(devs), arr, _lib.GROUP_REHEARSAL if rehearsal else 0, C.byref(self._g))
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        check(rc, "rrtx_group_create")
        self.stats = None
        self._scene = None

    def close(self):
        g = getattr(self, "_g", None)
        if g:
            lib.rrtx_group_destroy(g)
            self._g = None

    __del__ = close

    def __len__(self):
        return lib.rrtx_group_size(self._g)

    def member_rows(self, i):
        ctx = C.c_void_p(lib.rrtx_group_member(self._g, i))
        n = lib.rrtx_shard_rows(ctx, None, 0)
        rows = (C.c_int32 * max(n, 1))()
        lib.rrtx_shard_rows(ctx, rows, n)
        return np.array(rows[:n], dtype=np.int32)

    def set_scene(self, scene):
        check(lib.rrtx_group_set_scene(self._g, C.byref(scene.desc)), "rrtx_group_set_scene")
        self._scene = scene

    def render(self, scene=None):
        if scene is not None:
            self.set_scene(scene)
        p = self.params
        fb = np.zeros((p.image_height, p.image_width, 3), dtype=np.float64 if self.fp64 else np.float32)
        st = _lib.GroupStats()
        check(lib.rrtx_group_render(self._g, fb.ctypes.data_as(C.c_void_p), C.byref(st)), "rrtx_group_render")
        self.stats = st.as_dict()
        return fb

    def render_device(self):
        """Render and leave the assembled frame on the first device; returns its device address."""
        ptr = C.c_void_p()
        st = _lib.GroupStats()
        check(lib.rrtx_group_render_device(self._g, C.byref(ptr), C.byref(st)), "rrtx_group_render_device")
        self.stats = st.as_dict()
        return ptr.value


def device_count():
    return lib.rrtx_device_count()


def query_device(device=0):
    info = _lib.DevInfo()
    check(lib.rrtx_query(device, C.byref(info)), "rrtx_query")
    return {n: (getattr(info, n).decode() if n == "name" else getattr(info, n)) for n, _ in info._fields_}


def quantise(fb, samples_per_pixel):
    """convert_color over the frame + row flip (color.h:8-23, main.cpp:153): -> uint8 [h, w, 3], top row first."""
    fb = np.ascontiguousarray(fb)
    if fb.dtype not in (np.float32, np.float64):
        raise TypeError("framebuffer must be float32 or float64")
    h, w, _ = fb.shape
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    check(lib.rrtx_quantise(fb.ctypes.data_as(C.c_void_p), int(fb.dtype == np.float64), w, h, int(samples_per_pixel), rgb.ctypes.data_as(C.c_void_p)), "rrtx_quantise")
    return rgb


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    check(lib.rrtx_write_ppm(path.encode() if path else None, rgb.ctypes.data_as(C.c_void_p), w, h), "rrtx_write_ppm")


def write_png(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    check(lib.rrtx_write_png(path.encode(), rgb.ctypes.data_as(C.c_void_p), w, h), "rrtx_write_png")
