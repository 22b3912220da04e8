"""One frame over several devices from ONE process (rrtx_group, include/rrtx.h): row-tile shards, one grouped RCCL
send / recv to the first device, de-interleave there - BASELINE.json's "row-tile-partitioned across the 8 GPUs of one
node with a final RCCL gather over xGMI", natively in the library and in `rrt -G`.

What a box with ONE GPU can prove, and does here:
  * N = 1 goes through the very code path N = 8 takes: ncclCommInitAll, ncclGroupStart, ncclSend + ncclRecv (rank 0
    sends its block to itself like every other rank), ncclGroupEnd, de-interleave - `stats["rccl"] == 1`;
  * N = device_count() members on distinct devices (= 1 here, 8 on a full node: the same test);
  * the N-way decomposition itself - shard geometry, block offsets, de-interleave for N = 2 ... 8 and several tile
    heights - as a REHEARSAL: the members share the device, their blocks move with device-to-device copies
    (RCCL cannot build a communicator over one device twice).
Every frame must equal the single-device render bit for bit.
"""
import os
import subprocess

import numpy as np
import pytest

from _oracle import GOLDEN, ROOT, Oracle, scene_path

pytestmark = pytest.mark.gpu

FINAL, TEST2, TEST3 = scene_path("final"), scene_path("test2"), scene_path("test3")


def _single(gpu, path, w, h, spp, fp64=False, use_bvh=False):
    r = gpu.Rrt(w, h, spp, 50, use_bvh=use_bvh, fp64=fp64)
    fb = r.render(gpu.Scene(path, w, h, fp64=fp64))
    st = r.stats
    r.close()
    return fb, st


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
def test_one_device_through_the_rccl_path(gpu, fp64):
    w, h, spp = 200, 120, 12
    want, st1 = _single(gpu, FINAL, w, h, spp, fp64=fp64)
    g = gpu.RrtGroup(1, w, h, spp, 50, use_bvh=False, fp64=fp64)
    fb = g.render(gpu.Scene(FINAL, w, h, fp64=fp64))
    st = g.stats
    assert st["rccl"] == 1 and st["n_devices"] == 1
    assert np.array_equal(fb, want)
    assert st["segments"] == st1["segments"] and st["samples"] == w * h * spp
    assert st["gathered_bytes"] == w * h * 3 * (8 if fp64 else 4)
    assert st["device_ms"] >= st["render_ms"] > 0 and st["gather_ms"] >= 0
    fb2 = g.render()  # a second frame on the live communicator
    assert np.array_equal(fb2, want)
    g.close()
    assert np.array_equal(want[5], Oracle(FINAL, w, h, fp64).render(spp, 50, 1984, order=1, chunk=8, rows=(5, 6))[0][5])


def test_rccl_does_not_talk_on_stdout(gpu):
    # stdout belongs to the PPM (main.cpp:142); RCCL prints a version banner there when a communicator is built
    import sys

    code = ("import sys; sys.path.insert(0, %r); import rrt_amd; g = rrt_amd.RrtGroup(1, 32, 20, 2, 5); "
            "g.render(rrt_amd.Scene(%r, 32, 20)); assert g.stats['rccl'] == 1; g.close()" % (ROOT, TEST3))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == b"", r.stdout


def test_every_device_of_the_box(gpu):
    n = gpu.device_count()
    w, h, spp = 160, 100, 6
    want, _ = _single(gpu, FINAL, w, h, spp, use_bvh=True)
    g = gpu.RrtGroup(n, w, h, spp, 50, use_bvh=True, tile_rows=4)
    assert len(g) == n
    fb = g.render(gpu.Scene(FINAL, w, h))
    st = g.stats
    # (on a box with one GPU this is N = 1; on the first lease with two or more it pins the multi-device exchange - ncclSend /
    # ncclRecv between distinct ranks, cross-device stream order - against the ORACLE without new code)
    assert st["rccl"] == 1 and st["n_devices"] == n == gpu.device_count() and st["accel_cells"] > 0
    assert st["rccl_comms"] == n and st["rccl_version"] > 0 and st["devices"] == list(range(n)) and st["accel_exact"] == 1
    assert np.array_equal(fb, want)
    fo, so = Oracle(FINAL, w, h, False).render(spp, 50, 1984, order=1, chunk=spp)
    assert np.array_equal(fb, fo) and st["segments"] == so["segments"]
    g.close()


@pytest.mark.parametrize("n,tile", [(2, 4), (3, 1), (4, 8), (8, 4), (8, 3), (5, 16)])
def test_rehearsal_of_the_n_way_decomposition(gpu, n, tile):
    w, h, spp = 96, 75, 5  # 75 rows: ragged last tile, unequal shards
    for path, fp64 in ((FINAL, False), (TEST2, True)):
        want, st1 = _single(gpu, path, w, h, spp, fp64=fp64)
        g = gpu.RrtGroup([0] * n, w, h, spp, 50, use_bvh=False, fp64=fp64, tile_rows=tile, rehearsal=True)
        fb = g.render(gpu.Scene(path, w, h, fp64=fp64))
        assert g.stats["rccl"] == 0 and g.stats["n_devices"] == n
        assert np.array_equal(fb, want), (n, tile, fp64)
        assert g.stats["segments"] == st1["segments"]
        rows = np.concatenate([g.member_rows(i) for i in range(n)])
        assert np.array_equal(np.sort(rows), np.arange(h))
        g.close()


def test_more_members_than_tiles(gpu):
    # 6 rows in tiles of 4 = 2 tiles for 4 members: two of them render nothing and send nothing
    w, h, spp = 40, 6, 3
    want, _ = _single(gpu, TEST3, w, h, spp)
    g = gpu.RrtGroup([0, 0, 0, 0], w, h, spp, 50, use_bvh=False, tile_rows=4, rehearsal=True)
    fb = g.render(gpu.Scene(TEST3, w, h))
    assert np.array_equal(fb, want)
    assert [len(g.member_rows(i)) for i in range(4)] == [4, 2, 0, 0]
    g.close()


def test_group_arguments_fail_loudly(gpu):
    with pytest.raises(gpu.RrtxError) as e:
        gpu.RrtGroup([0, 0], 16, 16, 1, 5)  # a device twice without the rehearsal flag
    assert e.value.code == -1
    with pytest.raises(gpu.RrtxError):
        gpu.RrtGroup([1000], 16, 16, 1, 5)
    with pytest.raises(gpu.RrtxError):
        gpu.RrtGroup(65, 16, 16, 1, 5)
    g = gpu.RrtGroup(1, 16, 16, 1, 5)
    with pytest.raises(gpu.RrtxError) as e:
        g.render()
    assert e.value.code == -3  # no scene
    g.close()


def test_config5_through_the_group(gpu):
    """BASELINE.json's 8-GPU configuration through rrtx_group: as a rehearsal of 8 members on this box's device
    (what the 8 GPUs of a node render, one after another - the streams of one device serialise) and through RCCL
    with the devices that exist.  At spp 100 (the full spp 1000 frame: tests/test_gpu_configs.py)."""
    w, h, spp = 3840, 2160, 100
    want, st1 = _single(gpu, FINAL, w, h, spp, use_bvh=True)
    sc = gpu.Scene(FINAL, w, h)
    g = gpu.RrtGroup([0] * 8, w, h, spp, 50, use_bvh=True, tile_rows=4, rehearsal=True)
    fb = g.render(sc)
    assert np.array_equal(fb, want)
    assert g.stats["gathered_bytes"] == w * h * 12 and g.stats["sample_chunk"] == st1["sample_chunk"]
    g.close()
    g = gpu.RrtGroup(gpu.device_count(), w, h, spp, 50, use_bvh=True, tile_rows=4)
    fb = g.render(sc)
    assert g.stats["rccl"] == 1 and np.array_equal(fb, want)
    g.close()


# ---- the CLI: rrt -G --------------------------------------------------------------------------------------------


def _cli(args):
    r = subprocess.run(args, capture_output=True, timeout=600)
    assert r.returncode == 0, (args, r.stderr[-3000:])
    return r


def test_cli_G_renders_one_frame_over_the_devices(gpu, tmp_path):
    from PIL import Image

    exe = os.path.join(ROOT, "rrt")
    w, h, spp = 150, 100, 6
    base = ["-i", FINAL, "-w", str(w), "-h", str(h), "-s", str(spp)]
    one = str(tmp_path / "one.png")
    _cli([exe] + base + ["-o", one])
    want = np.asarray(Image.open(one))
    # every device present, through RCCL
    n = gpu.device_count()
    out = str(tmp_path / "g.png")
    r = _cli([exe] + base + ["-o", out, "-G", str(n)] if n > 1 else [exe] + base + ["-o", out, "-G", "1"])
    assert np.array_equal(np.asarray(Image.open(out)), want)
    # 4 and 8 members sharing this box's device(s): the decomposition of BASELINE.json's 8-GPU run
    for members, tile in ((4, 4), (8, 2)):
        r = _cli([exe] + base + ["-o", out, "-G", str(members), "-E", "-T", str(tile)])
        assert b"gather" in r.stderr and b"took " in r.stderr and b"stats," in r.stderr
        assert np.array_equal(np.asarray(Image.open(out)), want), members
    # PPM on stdout, rrtd
    exed = os.path.join(ROOT, "rrtd")
    a = _cli([exed] + base)
    b = _cli([exed] + base + ["-G", "3", "-E"])
    assert a.stdout == b.stdout and a.stdout.startswith(b"P3\n")
    # more GPUs than the box has, without -E: a device error, like a bad -D
    r = subprocess.run([exe] + base + ["-o", out, "-G", str(n + 1)], capture_output=True)
    assert r.returncode == 99


def test_cli_G_deals_a_batch_of_scenes_to_the_devices(gpu, tmp_path):
    """frame-level data parallelism - with at least as many scenes as -G members every member
    renders whole frames on its own device, with its own writer task; every image equals the one a single process
    per scene writes, and PPMs on stdout keep the order of the command line."""
    from PIL import Image

    exe = os.path.join(ROOT, "rrt")
    w, h, spp = 120, 80, 5
    xform = os.path.join(GOLDEN, "scenes", "xform.txt")
    names = [FINAL, TEST2, TEST3, xform, FINAL, scene_path("test1"), TEST2, FINAL, TEST3]
    single = {}
    for s in set(names):
        o = str(tmp_path / "s.png")
        _cli([exe, "-i", s, "-o", o, "-w", str(w), "-h", str(h), "-s", str(spp)])
        single[s] = np.asarray(Image.open(o)).copy()
    for members in (gpu.device_count(), 2, 4):
        outs = [str(tmp_path / ("b%d_%d.png" % (members, i))) for i in range(len(names))]
        args = [exe, "-w", str(w), "-h", str(h), "-s", str(spp), "-G", str(members)] + ([] if members <= gpu.device_count() else ["-E"])
        for s, o in zip(names, outs):
            args += ["-i", s, "-o", o]
        r = _cli(args)
        assert r.stderr.count(b"took ") == len(names) and r.stdout == b""
        for s, o in zip(names, outs):
            assert np.array_equal(np.asarray(Image.open(o)), single[s]), (members, s)
    # PPMs of a dealt-out batch: in command-line order on stdout
    args = [exe, "-w", "40", "-h", "30", "-s", "3", "-G", "3", "-E"]
    for s in names[:6]:
        args += ["-i", s]
    r = _cli(args)
    seq = b""
    for s in names[:6]:
        seq += _cli([exe, "-w", "40", "-h", "30", "-s", "3", "-i", s]).stdout
    assert r.stdout == seq
    # a scene that does not parse: the reference's exit code, nothing after it is started
    bad = tmp_path / "bad.txt"
    bad.write_text("camera 0 0 1 0 0 0 0 1 0 40 0 1\nmaterial a nosuchmaterial 1 1 1\nsphere 0 0 0 1 a\n")
    r = subprocess.run([exe, "-w", "40", "-h", "30", "-s", "1", "-G", "2", "-E", "-i", str(bad), "-o", str(tmp_path / "x.png"), "-i", FINAL, "-o", str(tmp_path / "y.png")], capture_output=True)
    assert r.returncode == 3
