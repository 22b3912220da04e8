"""ctypes binding of the C ABI declared in include/rrtx.h (rrt_amd/librrtx.so).

There is no Python or CPU fallback: if the HIP library has not been built, importing this
module raises, and every render entry point needs a visible MI355X.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrtx.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rrtx.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "rrt_amd: %s is missing - build it with `make` (or __graft_entry__.build()); this package has no fallback path" % LIB_PATH
    )

# PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one process do
# not coexist ("No HIP GPUs are available" from whichever initialises second), so when torch is
# installed load it FIRST: librrtx.so's NEEDED libamdhip64.so.7 then binds to the copy torch already
# mapped and the whole process shares one runtime (and one set of streams, which is what lets
# rrtx_render_device() launch on torch's current stream).  Without torch (the rrt / rrtd binaries,
# plain ctypes users) the system ROCm runtime is used.
try:  # pragma: no cover - depends on the environment
    import torch  # noqa: F401
except Exception:  # torch absent or broken: fall through to the system runtime
    torch = None

lib = C.CDLL(LIB_PATH)


class Params(C.Structure):  # rrtx_params
    _fields_ = [
        ("image_width", C.c_int32),
        ("image_height", C.c_int32),
        ("samples_per_pixel", C.c_int32),
        ("max_depth", C.c_int32),
        ("use_bvh", C.c_int32),
        ("threads_x", C.c_int32),
        ("threads_y", C.c_int32),
        ("fp64", C.c_int32),
        ("device", C.c_int32),
        ("seed", C.c_uint32),
        ("sample_chunk", C.c_int32),
        ("shard_rank", C.c_int32),
        ("shard_count", C.c_int32),
        ("tile_rows", C.c_int32),
        ("collect_stats", C.c_int32),
        ("flags", C.c_int32),
        ("handoff_lanes", C.c_int32),
        ("handoff_iters", C.c_int32),
        ("list_passes", C.c_int32),
        ("taper_samples", C.c_int32),
        ("reserved", C.c_int32 * 1),
    ]


class Stats(C.Structure):  # rrtx_stats
    _fields_ = [
        ("kernel_ms", C.c_double),
        ("kernel_ms_sum", C.c_double),
        ("renders", C.c_int32),
        ("accel_cells", C.c_int32),
        ("wall_ms", C.c_double),
        ("samples", C.c_uint64),
        ("segments", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("bytes_algorithmic", C.c_uint64),
        ("grid_blocks", C.c_int32),
        ("block_threads", C.c_int32),
        ("sample_chunk", C.c_int32),
        ("local_rows", C.c_int32),
        ("candidates", C.c_uint64),
        ("scan_filter", C.c_int32),
        ("list_mismatches", C.c_int32),
        ("scanned_segments", C.c_uint64),
        ("accel_exact", C.c_int32),
        ("scan_mfma", C.c_int32),
        ("walk_cells", C.c_uint64),
        ("walk_pairs", C.c_uint64),
        ("convergence_faults", C.c_uint64),
        ("sky_pixels", C.c_int32),
        ("first_bounce", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


FLAG_SCAN_SCALAR_ONLY, FLAG_SCAN_LDS_ONLY, FLAG_NO_TAIL_KERNEL, FLAG_NO_PRIMARY_LISTS, FLAG_VERIFY_LISTS = 2, 4, 8, 16, 32


class SceneDesc(C.Structure):  # rrtx_scene_desc
    _fields_ = [
        ("fp64", C.c_int32),
        ("camera", C.c_void_p),
        ("num_materials", C.c_int32),
        ("materials", C.c_void_p),
        ("num_spheres", C.c_int32),
        ("spheres", C.c_void_p),
        ("num_moving_spheres", C.c_int32),
        ("moving_spheres", C.c_void_p),
        ("num_triangles", C.c_int32),
        ("triangles", C.c_void_p),
    ]


class DevInfo(C.Structure):  # rrtx_devinfo
    _fields_ = [
        ("name", C.c_char * 256),
        ("major", C.c_int32),
        ("minor", C.c_int32),
        ("multi_processor_count", C.c_int32),
        ("shared_mem_per_block", C.c_int64),
        ("max_threads_per_block", C.c_int32),
        ("max_threads_per_multiprocessor", C.c_int32),
        ("unified_addressing", C.c_int32),
        ("l2_cache_size", C.c_int32),
        ("total_global_mem", C.c_int64),
        ("clock_khz", C.c_int32),
    ]


class GroupStats(C.Structure):  # rrtx_group_stats
    _fields_ = [
        ("n_devices", C.c_int32),
        ("rccl", C.c_int32),
        ("render_ms", C.c_double),
        ("device_ms", C.c_double),
        ("gather_ms", C.c_double),
        ("wall_ms", C.c_double),
        ("kernel_ms", C.c_double * 16),
        ("samples", C.c_uint64),
        ("segments", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("bytes_algorithmic", C.c_uint64),
        ("gathered_bytes", C.c_uint64),
        ("sample_chunk", C.c_int32),
        ("accel_cells", C.c_int32),
        ("accel_exact", C.c_int32),
        ("rccl_version", C.c_int32),
        ("rccl_comms", C.c_int32),
        ("devices", C.c_int32 * 16),
        ("reserved", C.c_int32),
    ]

    def as_dict(self):
        n = max(1, min(16, self.n_devices))
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("kernel_ms", "devices", "reserved")}
        d["kernel_ms"] = list(self.kernel_ms)[:n]
        d["devices"] = list(self.devices)[:n]
        return d


GROUP_REHEARSAL = 1
GROUP_KEEP_STDOUT = 2


def _sig(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


_sig("rrtx_version", C.c_char_p, [])
_sig("rrtx_abi_version", C.c_int, [])
ABI_VERSION = 4  # RRTX_ABI_VERSION of include/rrtx.h as mirrored by the Structures above
if lib.rrtx_abi_version() != ABI_VERSION:
    raise ImportError("rrt_amd: %s was built with RRTX_ABI_VERSION %d, this binding mirrors %d - rebuild with `make`" % (LIB_PATH, lib.rrtx_abi_version(), ABI_VERSION))
_sig("rrtx_last_error", C.c_char_p, [])
_sig("rrtx_device_count", C.c_int, [])
_sig("rrtx_query", C.c_int, [C.c_int, C.POINTER(DevInfo)])
_sig("rrtx_runtime_version", C.c_int, [])
_sig("rrtx_pin_host", C.c_int, [C.c_void_p, C.c_size_t])
_sig("rrtx_unpin_host", C.c_int, [C.c_void_p])
_sig("rrtx_create", C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)])
_sig("rrtx_destroy", None, [C.c_void_p])
_sig("rrtx_set_scene", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)])
_sig("rrtx_shard_rows", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int])
_sig("rrtx_render", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Stats)])
_sig("rrtx_render_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p])
_sig("rrtx_collect", C.c_int, [C.c_void_p, C.POINTER(Stats)])
_sig("rrtx_stream", C.c_void_p, [C.c_void_p])
_sig("rrtx_group_create", C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_void_p)])
_sig("rrtx_group_destroy", None, [C.c_void_p])
_sig("rrtx_group_size", C.c_int, [C.c_void_p])
_sig("rrtx_group_member", C.c_void_p, [C.c_void_p, C.c_int])
_sig("rrtx_group_set_scene", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)])
_sig("rrtx_group_render", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(GroupStats)])
_sig("rrtx_group_render_device", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(GroupStats)])
_sig("rrtx_scene_load", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)])
_sig("rrtx_scene_exit_code", C.c_int, [])
_sig("rrtx_scene_error", C.c_char_p, [])
_sig("rrtx_scene_load_quiet", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)])
_sig("rrtx_scene_free", None, [C.c_void_p])
_sig("rrtx_scene_describe", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)])
_sig("rrtx_scene_counts", C.c_int, [C.c_void_p, C.POINTER(C.c_int32)])
_sig("rrtx_quantise", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p])
_sig("rrtx_write_ppm", C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int])
_sig("rrtx_write_png", C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int])


FLAG_EXACT_SCAN = 1
FLAG_EXACT_ACCEL = 64
FLAG_NO_TAIL_GRID = 128
FLAG_SCAN_NO_MFMA = 256
FLAG_ONE_ITEM_PER_PIXEL = 512
FLAG_NO_SKY_SPLIT = 1024
FLAG_NO_FIRST_BOUNCE = 2048
FLAG_FIRST_BOUNCE_ALWAYS = 4096
FLAG_SKY_SAME_STREAM = 8192


class RrtxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.rrtx_last_error().decode(errors="replace")
        super().__init__("%s failed with code %d: %s" % (where, code, msg))


def check(code, where):
    if code != 0:
        raise RrtxError(code, where)
