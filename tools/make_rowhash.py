"""Per-row hashes of the BASELINE frames at their STATED size, from the oracle (CPU restatement, pinned bit for bit by
the compiled reference: tests/test_oracle_vs_reference.py).  tests/test_gpu_configs.py compares EVERY hashed row of the
HIP frame with these, so the whole headline frame is pinned, not three rows of it.

    python tools/make_rowhash.py [c3] [c4] [c5]          (no argument: all three; ~6 / ~9 / ~5 minutes on 8 cores)

    c3  scenes/final.txt 1200x800  spp 500  fp32  chunk 8   -> tests/golden/c3_rowhash_f32.npy   uint64[800]
    c4  scenes/final.txt 1200x800  spp 500  fp64  chunk 8   -> tests/golden/c4_rowhash_f64.npy   uint64[800]
    c5  scenes/final.txt 3840x2160 spp 1000 fp32  chunk 16  -> tests/golden/c5_rowhash_f32.npz   rows int32[64], hash uint64[64]

The hash of a row is blake2b-64 over the row's raw radiance bytes ([W][3] floats / doubles as they lie in memory);
`chunk` is the summation shape rrtx_create chooses for that frame (rrtx_api.cpp, a function of W, H, spp only).  The
segment counts of the rendered rows go to tests/golden/rowhash_meta.json.  The fixtures are DATA: hashes of outputs.
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from _oracle import GOLDEN, Oracle, scene_path

FINAL = scene_path("final")
C5_ROWS = [int(round(k * 2159 / 63)) for k in range(64)]  # 64 rows from the ground to the sky, first and last included


def row_hash(row):
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(row).tobytes(), digest_size=8).digest(), "little")


def render_rows(o, spp, chunk, rows, block):
    """-> {row: hash}, total segments; renders `block` consecutive rows per oracle call where the rows are consecutive"""
    out, seg, t0 = {}, 0, time.time()
    k = 0
    while k < len(rows):
        e = k + 1
        while e < len(rows) and e - k < block and rows[e] == rows[e - 1] + 1:
            e += 1
        fb, st = o.render(spp, 50, 1984, order=1, chunk=chunk, rows=(rows[k], rows[e - 1] + 1))
        for j in rows[k:e]:
            out[j] = row_hash(fb[j])
        seg += st["segments"]
        k = e
        print("  %d / %d rows, %.0f s" % (k, len(rows), time.time() - t0), flush=True)
    return out, seg


def main(which):
    meta_path = os.path.join(GOLDEN, "rowhash_meta.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    if "c3" in which:
        h, seg = render_rows(Oracle(FINAL, 1200, 800, False), 500, 8, list(range(800)), 50)
        np.save(os.path.join(GOLDEN, "c3_rowhash_f32.npy"), np.array([h[j] for j in range(800)], dtype=np.uint64))
        meta["c3"] = {"w": 1200, "h": 800, "spp": 500, "chunk": 8, "fp64": False, "rows": 800, "segments": seg}
    if "c4" in which:
        h, seg = render_rows(Oracle(FINAL, 1200, 800, True), 500, 8, list(range(800)), 50)
        np.save(os.path.join(GOLDEN, "c4_rowhash_f64.npy"), np.array([h[j] for j in range(800)], dtype=np.uint64))
        meta["c4"] = {"w": 1200, "h": 800, "spp": 500, "chunk": 8, "fp64": True, "rows": 800, "segments": seg}
    if "c5" in which:
        h, seg = render_rows(Oracle(FINAL, 3840, 2160, False), 1000, 16, C5_ROWS, 1)
        np.savez(os.path.join(GOLDEN, "c5_rowhash_f32.npz"), rows=np.array(C5_ROWS, dtype=np.int32), hash=np.array([h[j] for j in C5_ROWS], dtype=np.uint64))
        meta["c5"] = {"w": 3840, "h": 2160, "spp": 1000, "chunk": 16, "fp64": False, "rows": len(C5_ROWS), "segments": seg}
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main([a.lower() for a in sys.argv[1:]] or ["c3", "c4", "c5"])
