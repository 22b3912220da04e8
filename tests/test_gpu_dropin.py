"""The drop-in seam proven by the reference's own `main`.

oracle/_ref/rrt_dropin (fp32) and rrtd_dropin (fp64) are the reference's unchanged main.cpp - its argv loop,
its scene.h parser, its color.h quantiser, its stb PNG writer - linked with oracle/ref_dropin.cpp, which
implements class Rrt (rrt.h:14-48) on top of librrtx.so exactly as INTEGRATION.md section 1 tells a maintainer
to.  They are built in the container by oracle/Makefile from the sources under /root/reference and travel to
the GPU box as prebuilt files.  Here they must produce, for every shipped scene, the PPM bytes and the PNG
pixels that the product's own `rrt` / `rrtd` produce: the reference's parser against ours, its quantiser
against ours, its table marshalling against rrtx_scene_describe(), both through the same HIP kernels.
"""
import os
import subprocess

import numpy as np
import pytest

from _oracle import GOLDEN, REF_DIR, ROOT, Oracle, scene_path

pytestmark = pytest.mark.gpu

SCENES = {"test1": scene_path("test1"), "test2": scene_path("test2"), "test3": scene_path("test3"), "final": scene_path("final"), "xform": os.path.join(GOLDEN, "scenes", "xform.txt")}


def _need(exe):
    path = os.path.join(REF_DIR, exe)
    assert os.path.exists(path), "%s missing: run `make -C oracle ref` in the container (needs /root/reference)" % path
    return path


def _run(exe, scene, w, h, spp, extra=(), png=None):
    args = [exe, "-i", scene, "-w", str(w), "-h", str(h), "-s", str(spp)] + list(extra)
    if png:
        args += ["-o", png]
    r = subprocess.run(args, capture_output=True, timeout=600)
    assert r.returncode == 0, (args, r.stderr[-2000:])
    return r


@pytest.mark.parametrize("fp64", [False, True], ids=["rrt", "rrtd"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_reference_main_over_librrtx_writes_what_rrt_writes(gpu, tmp_path, name, fp64):
    from PIL import Image

    dropin = _need("rrtd_dropin" if fp64 else "rrt_dropin")
    ours = os.path.join(ROOT, "rrtd" if fp64 else "rrt")
    w, h, spp = (400, 266, 4) if name == "test1" else (150, 100, 6)  # test1: BASELINE.json's configuration 1
    for extra in ([], ["-b"]):  # default = accelerated closest hit, -b = list scan (main.cpp:67,90)
        a = _run(dropin, SCENES[name], w, h, spp, extra)
        b = _run(ours, SCENES[name], w, h, spp, extra)
        assert a.stdout == b.stdout and a.stdout.startswith(b"P3\n%d %d\n255\n" % (w, h)), (name, extra)  # PPM: byte for byte
        assert b"took " in a.stderr
        pa, pb = str(tmp_path / "a.png"), str(tmp_path / "b.png")
        a = _run(dropin, SCENES[name], w, h, spp, extra, png=pa)
        b = _run(ours, SCENES[name], w, h, spp, extra, png=pb)
        assert a.stdout == b"" and b.stdout == b""
        ia, ib = np.asarray(Image.open(pa)), np.asarray(Image.open(pb))  # (stb's deflate and ours differ: pixels, not file bytes)
        assert ia.shape == (h, w, 3) and np.array_equal(ia, ib), (name, extra)
    # ... and both equal the oracle's image
    fo, _ = Oracle(SCENES[name], w, h, fp64).render(spp, 50, 1984, order=1, chunk=spp)
    assert np.array_equal(ia, Oracle.quantise(fo, spp)), name


def test_reference_main_keeps_its_exit_codes_over_librrtx(gpu, tmp_path):
    dropin = _need("rrt_dropin")
    r = subprocess.run([dropin], capture_output=True)
    assert r.returncode == 1 and b"ERROR: no scene loaded." in r.stderr  # main.cpp:126-128
    r = subprocess.run([dropin, "-x"], capture_output=True)
    assert r.returncode == 1 and b"Unexpected argument" in r.stderr  # main.cpp:33-50
    bad = tmp_path / "bad.txt"
    bad.write_text("camera 0 0 1 0 0 0 0 1 0 40 0 1\nmaterial a lambertian 1 1 1\n")  # no primitives: scene.h:437-441, exit 4
    r = subprocess.run([dropin, "-i", str(bad)], capture_output=True)
    assert r.returncode == 4
