"""rrt_amd - host-side mirror of the rogerallen/rrt render interface over the MI355X-native C ABI.

    scene = rrt_amd.Scene("scenes/final.txt", 1200, 800)          # scene.h:212  scene(filename, w, h)
    rrt   = rrt_amd.Rrt(1200, 800, 500, 50, use_bvh=True)         # rrt.h:16     Rrt(w, h, spp, depth, bvh[, tx, ty])
    fb    = rrt.render(scene)                                     # rrt.h:34     vec3* render(scene*)
    rgb   = rrt_amd.quantise(fb, 500)                             # color.h:8    convert_color, main.cpp:153 row flip
    rrt_amd.write_png("out.png", rgb)                             # main.cpp:164

Everything computes in librrtx.so (hand-written HIP for gfx950); nothing here falls back to
Python/CPU arithmetic.
"""
from ._lib import RrtxError  # noqa: F401
from ._lib import FLAG_NO_FIRST_BOUNCE, FLAG_SKY_SAME_STREAM, FLAG_FIRST_BOUNCE_ALWAYS, FLAG_EXACT_ACCEL, FLAG_EXACT_SCAN, FLAG_NO_SKY_SPLIT, FLAG_NO_TAIL_GRID, FLAG_ONE_ITEM_PER_PIXEL, FLAG_SCAN_NO_MFMA  # noqa: F401  (rrtx_params.flags, include/rrtx.h)
from .render import Rrt, RrtGroup, Scene, device_count, query_device, quantise, write_png, write_ppm  # noqa: F401

__all__ = ["Rrt", "RrtGroup", "Scene", "RrtxError", "FLAG_EXACT_SCAN", "FLAG_EXACT_ACCEL", "FLAG_NO_TAIL_GRID", "FLAG_SCAN_NO_MFMA", "FLAG_ONE_ITEM_PER_PIXEL", "FLAG_NO_SKY_SPLIT", "FLAG_NO_FIRST_BOUNCE", "FLAG_SKY_SAME_STREAM", "FLAG_FIRST_BOUNCE_ALWAYS", "device_count", "query_device", "quantise", "write_png", "write_ppm"]
