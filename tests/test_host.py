"""Host side of the seam (no GPU needed): scene parser, quantiser, PPM/PNG writers and the CLI's
argument handling, against the committed reference outputs (tests/golden) and the oracle."""
import os
import subprocess

import numpy as np
import pytest

import rrt_amd
from _oracle import GOLDEN, ROOT, Oracle, scene_path

SCENES = {"test1": scene_path("test1"), "test2": scene_path("test2"), "test3": scene_path("test3"), "final": scene_path("final"), "xform": os.path.join(GOLDEN, "scenes", "xform.txt")}
W, H, SPP = 32, 20, 3


def _flat_tables(t):
    cam = np.concatenate([t["camera"]["v"].reshape(-1), [t["camera"]["lens_radius"], t["camera"]["time0"], t["camera"]["time1"]]]).astype(np.float64)
    m = t["materials"]
    mats = np.zeros((len(m), 6))
    mats[:, 0] = m["type"]
    for i, ty in enumerate(m["type"]):
        if ty in (0, 1):
            mats[i, 1:4] = m["albedo"][i]
        if ty == 1:
            mats[i, 4] = m["fuzz"][i]
        if ty == 2:
            mats[i, 5] = m["ref_idx"][i]
    s = t["spheres"]
    sph = np.column_stack([s["center"].astype(np.float64).reshape(len(s), 3), s["radius"], s["material_idx"]]) if len(s) else np.zeros((0, 5))
    ms = t["moving_spheres"]
    msph = np.column_stack([ms["center0"].astype(np.float64).reshape(len(ms), 3), ms["center1"].astype(np.float64).reshape(len(ms), 3), ms["time0"], ms["time1"], ms["radius"], ms["material_idx"]]) if len(ms) else np.zeros((0, 10))
    tr = t["triangles"]
    tris = np.column_stack([tr["vertices"].astype(np.float64).reshape(len(tr), 9), tr["material_idx"]]) if len(tr) else np.zeros((0, 10))
    return dict(cam=cam, materials=mats, spheres=sph, msph=msph, tris=tris)


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_scene_parser_matches_reference_tables(name, fp64):
    g = np.load(os.path.join(GOLDEN, "tables_%s_%s.npz" % (name, "f64" if fp64 else "f32")))
    sc = rrt_amd.Scene(SCENES[name], W, H, fp64=fp64)
    assert sc.counts() == list(g["counts"])
    flat = _flat_tables(sc.tables())
    for key in flat:
        assert np.array_equal(flat[key], g[key]), key


def test_table_struct_sizes_match_reference():
    import rrt_amd.render as rr

    for fp64 in (False, True):
        sz = list(np.load(os.path.join(GOLDEN, "tables_test1_%s.npz" % ("f64" if fp64 else "f32")))["sizeof"])
        dt = rr._table_dtypes(fp64)
        assert [dt["camera"].itemsize, dt["material"].itemsize, dt["sphere"].itemsize, dt["msphere"].itemsize, dt["triangle"].itemsize] == sz[1:]


def test_scene_errors_carry_reference_exit_codes(tmp_path):
    def code(text):
        p = tmp_path / "s.txt"
        p.write_text(text)
        try:
            rrt_amd.Scene(str(p), 8, 8)
        except ValueError as e:
            return int(str(e).split()[-1].rstrip(")"))
        return 0

    cam = "camera 0 0 5 0 0 0 0 1 0 30 0.1 5\n"
    mat = "material m lambertian 0.5 0.5 0.5\n"
    assert code(cam + mat + "sphere 0 0 0 1 m\n") == 0
    assert code(cam + mat + "sphere 0 0 0 1 m\nobj 0\n") == 1  # instance of an undefined object
    assert code(mat + "sphere 0 0 0 1 m\n") == 4
    assert code(cam + "sphere 0 0 0 1 m\n") == 4
    assert code(cam + mat) == 4
    assert code(cam + "material m plastic 1 1 1\n") == 3
    assert code(cam + mat + "obj_vtx 0 0 0\n") == 1
    assert code(cam + mat + "obj_beg 1 1\nobj_vtx 0 0 0\nobj_end\n") == 1
    assert code(cam + mat + "obj_beg 1 0\nobj_vtx 0 0 0\nobj_vtx 1 0 0\n") == 1
    with pytest.raises(ValueError) as e:
        rrt_amd.Scene(str(tmp_path / "missing.txt"), 8, 8)
    assert "exit code 2" in str(e.value)


def test_ignored_lines_and_name_rules(tmp_path):
    # leading whitespace / comments are ignored; first definition of a material name wins; an unknown
    # name maps to material 0 (scene.h:310)
    p = tmp_path / "s.txt"
    p.write_text("camera 0 0 5 0 0 0 0 1 0 30 0.1 5\n  sphere 9 9 9 1 a\n#sphere 9 9 9 1 a\nmaterial a lambertian 1 0 0\nmaterial b metal 0 1 0 0.5\n"
                 "material a dielectric 1.5\nsphere 0 0 0 1 a\nsphere 1 0 0 1 b\nsphere 2 0 0 1 zzz\nsphere 3 0 0 1\n")
    t = rrt_amd.Scene(str(p), 8, 8).tables()
    assert list(t["materials"]["type"]) == [0, 1, 2]
    assert list(t["spheres"]["material_idx"]) == [0, 1, 0, 0]
    o = Oracle(str(p), 8, 8).tables()
    assert list(o.spheres[:, 4]) == [0, 1, 0, 0]


def test_quantiser_matches_reference_cases():
    cases = np.load(os.path.join(GOLDEN, "quantise_cases.npy"))
    for fp64 in (0, 1):
        for spp in np.unique(cases[:, 1]):
            sel = cases[(cases[:, 0] == fp64) & (cases[:, 1] == spp)]
            fb = sel[:, 2:5].astype(np.float64 if fp64 else np.float32).reshape(1, -1, 3)
            got = rrt_amd.quantise(fb, int(spp))[0]
            want = (sel[:, 5:8].astype(np.int64) & 0xFF).astype(np.uint8)
            assert np.array_equal(got, want), (fp64, spp)


def test_quantiser_nan_and_negative():
    fb = np.array([[[np.nan, -1.0, -0.0], [np.inf, 1e-45, 4.0]]], dtype=np.float32)
    got = rrt_amd.quantise(fb, 4)
    assert np.array_equal(got, Oracle.quantise(fb, 4))
    assert list(got[0, 0]) == [0, 0, 0] and list(got[0, 1]) == [255, 0, 255]


def test_quantised_frame_and_row_flip_match_reference():
    fb = np.load(os.path.join(GOLDEN, "radiance_test1_f32.npy"))
    assert np.array_equal(rrt_amd.quantise(fb, SPP), np.load(os.path.join(GOLDEN, "frame_test1_f32_rgb.npy")))
    fb = np.load(os.path.join(GOLDEN, "radiance_final_f64.npy"))
    assert np.array_equal(rrt_amd.quantise(fb, SPP), np.load(os.path.join(GOLDEN, "frame_final_f64_rgb.npy")))


def test_ppm_text_is_byte_identical_to_reference(tmp_path):
    rgb = np.load(os.path.join(GOLDEN, "frame_test1_f32_rgb.npy"))
    out = tmp_path / "o.ppm"
    rrt_amd.write_ppm(str(out), rgb)
    assert out.read_bytes() == open(os.path.join(GOLDEN, "frame_test1_f32.ppm"), "rb").read()


def test_png_decodes_to_the_same_pixels(tmp_path):
    from PIL import Image

    rgb = np.load(os.path.join(GOLDEN, "frame_final_f64_rgb.npy"))
    out = tmp_path / "o.png"
    rrt_amd.write_png(str(out), rgb)
    img = Image.open(str(out))
    assert img.mode == "RGB" and img.size == (rgb.shape[1], rgb.shape[0])
    assert np.array_equal(np.asarray(img), rgb)
    noise = np.random.default_rng(0).integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    rrt_amd.write_png(str(out), noise)
    assert np.array_equal(np.asarray(Image.open(str(out))), noise)


@pytest.mark.parametrize("shape", [(1, 1), (3, 4000), (2160, 3840), (1001, 777), (5000, 2)])
def test_png_bands_join_into_one_valid_stream(tmp_path, shape):
    """Large images are deflated in bands of rows on several threads and joined behind one zlib header
    (host_image.cpp); PIL, i.e. zlib's inflate with its Adler-32 check, must decode exactly the input:
    noise (incompressible), a smooth gradient and rows of constants, for frames with fewer rows than
    threads, one pixel, and a 4K frame."""
    from PIL import Image

    h, w = shape
    rng = np.random.default_rng(h * 7919 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    for k, img in enumerate((rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8),
                             np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], axis=-1).astype(np.uint8),
                             np.full((h, w, 3), 200, dtype=np.uint8))):
        out = tmp_path / ("b%d.png" % k)
        rrt_amd.write_png(str(out), np.ascontiguousarray(img))
        back = Image.open(str(out))
        back.load()  # (decodes and verifies the checksums)
        assert back.size == (w, h) and np.array_equal(np.asarray(back), img), (shape, k)


@pytest.mark.parametrize("binary", ["rrt", "rrtd"])
def test_cli_argument_errors_exit_like_the_reference(binary, tmp_path):
    exe = os.path.join(ROOT, binary)
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 1 and b"ERROR: no scene loaded." in r.stderr and r.stdout == b""
    r = subprocess.run([exe, "-z"], capture_output=True)
    assert r.returncode == 1 and b"Unexpected argument: -z" in r.stderr and b"Usage: rrt [options]" in r.stderr
    r = subprocess.run([exe, "stray"], capture_output=True)
    assert r.returncode == 1 and b"Unexpected argument: stray" in r.stderr
    r = subprocess.run([exe, "-i", str(tmp_path / "nope.txt")], capture_output=True)
    assert r.returncode == 2 and b"problem with opening file" in r.stderr
    bad = tmp_path / "bad.txt"
    bad.write_text("camera 0 0 5 0 0 0 0 1 0 30 0.1 5\nmaterial m plastic 1 1 1\n")
    r = subprocess.run([exe, "-i", str(bad)], capture_output=True)
    assert r.returncode == 3 and b"unknown material type: plastic" in r.stderr


def test_fast_division_by_launch_constants_is_exact(tmp_path):
    """task_decode divides by W, tile_rows, spp and chunks_per_pixel with a multiply-high (rrtx_device.h
    make_fastdiv); the host check compares it with `/` for n < 2^31 over ~100M (n, d) pairs."""
    exe = tmp_path / "fastdiv_check"
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "fastdiv_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert " bad 0" in out, out


def test_grid_walk_equals_the_sequential_scan_on_the_host(tmp_path):
    """The accelerated closest hit (use_bvh) — grid builder, exact tests, tie rules, DDA walk in one go and
    in slices — compiled for the CPU from the headers the kernel is built from, against the sequential
    hittable_list scan on ~1.2 M rays (camera, surface, inside-sphere, lattice-aligned, grazing, far-away
    origins; coincident / nested / tiny / moving spheres, triangles), fp32 and fp64 — and in fp64 on a mesh
    of 4000 gridded triangles with rays in and near the triangles' planes; the same mesh
    in fp32 must NOT get a grid (the bound admits no triangle there).  Since round 3 every ray also goes through the BATCHED walk
    (accel_closest_hit_batched: what the densely pairing kernels run - cells listed first, entries tested afterwards in any order,
    with the empty-block skipping on every grid: -DRRTX_GRID_COARSE_ALWAYS=1), and a fourth scene holds piles of 300 and 1 300
    spheres in a cell or two (cells that take several ranges, cells their lane tests alone)."""
    exe = tmp_path / "path_host_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-DRRTX_GRID_COARSE_ALWAYS=1", os.path.join(ROOT, "tests", "path_host_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "400000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 mismatches, 0 sliced-walk mismatches, 0 batched-walk mismatches") == 7, r.stdout
    assert "fp32 variant 3" in r.stdout and "fp64 variant 3" in r.stdout
    assert "fp64 variant 2" in r.stdout and "4 always" in r.stdout.split("fp64 variant 2")[1], r.stdout


@pytest.fixture(scope="module")
def host_render(tmp_path_factory):
    """tests/path_host_render.cpp: the kernel's path arithmetic (rrtx_path.h) compiled for the host."""
    exe = tmp_path_factory.mktemp("hostrender") / "path_host_render"
    lib_dir = os.path.join(ROOT, "rrt_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "path_host_render.cpp"), "-o", str(exe), "-L" + lib_dir, "-lrrtx",
                    "-Wl,-rpath," + lib_dir], check=True)

    def run(scene, w, h, spp, depth, fp64, chunk, mode):
        out = str(exe) + ".raw"
        r = subprocess.run([str(exe), scene, str(w), str(h), str(spp), str(depth), str(int(fp64)), str(chunk), str(mode), out], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        info = dict(zip(r.stdout.split()[0::2], (int(x) for x in r.stdout.split()[1::2])))
        return np.fromfile(out, dtype=np.float64 if fp64 else np.float32).reshape(h, w, 3), info

    return run


@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
@pytest.mark.parametrize("name", ["test1", "test2", "test3", "final", "xform"])
def test_kernel_source_compiled_for_the_host_equals_the_oracle(host_render, name, fp64):
    """The product's own path arithmetic — the headers rrtx_kernels.hip is built from — run on the CPU by a
    plain loop, against the oracle: bit-identical radiance for every scene, in the list-scan and the
    grid-walk formulation of the closest hit, with chunked and whole-pixel sums.  (What the GPU tests
    then add is the kernel's scheduling around this arithmetic.)"""
    path = os.path.join(GOLDEN, "scenes", "xform.txt") if name == "xform" else scene_path(name)
    w, h, spp = 40, 26, 5
    for chunk in (8, 2):
        want, st = Oracle(path, w, h, fp64).render(spp, 50, 1984, order=1, chunk=min(chunk, spp))
        for mode in (0, 1):
            got, info = host_render(path, w, h, spp, 50, fp64, chunk, mode)
            assert info["segments"] == st["segments"], (chunk, mode)
            assert np.array_equal(got, want), (chunk, mode)
            if name == "final":
                assert (info["grid"] > 0) == (mode == 1) and (mode == 0 or info["walked"] > 0.99 * info["segments"])
    want, _ = Oracle(path, w, h, fp64).render(3, 2, 1984, order=1, chunk=3)  # depth limit
    assert np.array_equal(host_render(path, w, h, 3, 2, fp64, -1, 1)[0], want)


def test_mesh_scene_on_the_host_grid_in_fp64_list_in_fp32(host_render, tmp_path):
    """SURVEY.md 8(f) N2 on the CPU tier: a mesh of 672 triangles (obj instances) through the kernel's path
    arithmetic compiled for the host.  fp64: the grid is built, nearly every segment is walked, radiance
    bit-identical to the oracle; fp32: the residual bound admits no triangle, no grid, the list is
    scanned - and equals the oracle too."""
    from _oracle import mesh_scene

    f, n_tri = mesh_scene(tmp_path / "mesh.txt", 8, 16)
    w, h, spp = 36, 24, 3
    for fp64 in (True, False):
        want, st = Oracle(f, w, h, fp64).render(spp, 50, 1984, order=1, chunk=3)
        for mode in (0, 1):
            got, info = host_render(f, w, h, spp, 50, fp64, 3, mode)
            assert info["segments"] == st["segments"] and np.array_equal(got, want), (fp64, mode)
            if mode == 1:
                assert (info["grid"] > 0) == fp64, (fp64, info)
                assert not fp64 or info["walked"] > 0.9 * info["segments"]


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    """Sanitizers run on the CPU build only (the GPU pool has none): the scene parser on every scene
    and on malformed input, the quantiser and the PNG / PPM writers; and the grid builder + grid walk
    of the accelerated closest hit (the kernel's source, compiled for the host)."""
    san = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    csrc = os.path.join(ROOT, "rrt_amd", "csrc")
    exe = tmp_path / "host_asan"
    subprocess.run(["g++"] + san + [os.path.join(ROOT, "tests", "sanitize", "host_asan_main.cpp"), os.path.join(csrc, "host_scene.cpp"), os.path.join(csrc, "host_image.cpp"), "-o", str(exe), "-lz"],
                   check=True)
    bad = []
    for i, text in enumerate(["camera 1 2 3 0 0 0 0 1 0 40 0.1 10\nmaterial a lambertian 1 1 1\nsphere 0 0 0\n", "obj_beg 3 1\nobj_vtx 0 0 0\nobj_vtx 1 0 0\nobj_end\n", "material a wood 1 1 1\n",
                              "camera 1 2 3 0 0 0 0 1 0 40 0.1 10\nmaterial a lambertian 1 1 1\nobj 5 a t 1 2 3\n", ""]):
        f = tmp_path / ("bad%d.txt" % i)
        f.write_text(text)
        bad.append(str(f))
    scenes = [scene_path(n) for n in ("test1", "test2", "test3", "final")] + [os.path.join(GOLDEN, "scenes", "xform.txt")]
    r = subprocess.run([str(exe)] + scenes + bad + [str(tmp_path / "missing.txt")], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    assert r.stdout.count("rc=0") == 2 * len(scenes) and "png 0 ppm 0" in r.stdout
    exe2 = tmp_path / "path_check_asan"
    subprocess.run(["g++"] + san + [os.path.join(ROOT, "tests", "path_host_check.cpp"), "-o", str(exe2)], check=True)
    r = subprocess.run([str(exe2), "60000"], capture_output=True, text=True)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr


def test_cli_batch_logic_under_thread_sanitizer(tmp_path):
    """The CLI's host logic for batches (rrt_main.cpp: scenes dealt to workers, the next scene parsed ahead by a helper
    thread, two writer tasks per worker over three frame buffers, PPMs on stdout in command-line order) under
    ThreadSanitizer, with the device half of the C-ABI replaced by a stand-in that paints a pattern of the scene's counts
    (tests/sanitize/fake_device.cpp - nothing is rendered here): no race reported, every image in its own file and equal
    to the one a process of its own writes, stdout in order."""
    csrc = os.path.join(ROOT, "rrt_amd", "csrc")
    exe = str(tmp_path / "rrt_tsan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-o", exe, os.path.join(csrc, "rrt_main.cpp"), os.path.join(csrc, "host_scene.cpp"), os.path.join(csrc, "host_image.cpp"),
                    os.path.join(ROOT, "tests", "sanitize", "fake_device.cpp"), "-lz", "-lpthread"], check=True)
    xform = os.path.join(GOLDEN, "scenes", "xform.txt")
    names = [scene_path("final"), scene_path("test1"), scene_path("test2"), scene_path("test3"), xform, scene_path("final"), scene_path("test2"), scene_path("test1"), xform]
    size = ["-w", "96", "-h", "64", "-s", "3"]

    def run(args, **kw):
        r = subprocess.run([exe] + args, capture_output=True, timeout=600, **kw)
        assert r.returncode == 0, r.stderr[-3000:]
        assert b"ThreadSanitizer" not in r.stderr, r.stderr.decode()[-6000:]
        return r

    single = {}
    for s in set(names):
        o = str(tmp_path / "single.png")
        run(size + ["-i", s, "-o", o])
        single[s] = open(o, "rb").read()
    assert len(set(single.values())) == len(single)  # (the stand-in tells the scenes apart)
    for shape in (["-G", "3", "-E"], ["-G", "2"], []):
        outs = [str(tmp_path / ("b%d.png" % i)) for i in range(len(names))]
        args = size + shape
        for s, o in zip(names, outs):
            args += ["-i", s, "-o", o]
        r = run(args)
        assert r.stderr.count(b"took ") == len(names) and r.stdout == b""
        for s, o in zip(names, outs):
            assert open(o, "rb").read() == single[s], (shape, s)
            os.remove(o)
    # PPMs: stdout keeps the order of the command line
    small = ["-w", "40", "-h", "30", "-s", "2"]
    want = b"".join(run(small + ["-i", s]).stdout for s in names[:6])
    for shape in (["-G", "2", "-E"], ["-G", "3", "-E"], []):
        args = small + shape
        for s in names[:6]:
            args += ["-i", s]
        assert run(args).stdout == want, shape
    # one frame over a group; a scene that does not parse ends the batch with the reference's exit code
    run(small + ["-G", "4", "-E", "-i", names[0], "-o", str(tmp_path / "g.png")])
    bad = tmp_path / "bad.txt"
    bad.write_text("camera 0 0 1 0 0 0 0 1 0 40 0 1\nmaterial a nosuchmaterial 1 1 1\nsphere 0 0 0 1 a\n")
    # ... and the same sources under AddressSanitizer + UBSan: a dealt-out batch and a batch on one device
    exe_a = str(tmp_path / "rrt_asan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe_a, os.path.join(csrc, "rrt_main.cpp"), os.path.join(csrc, "host_scene.cpp"),
                    os.path.join(csrc, "host_image.cpp"), os.path.join(ROOT, "tests", "sanitize", "fake_device.cpp"), "-lz", "-lpthread"], check=True)
    for shape in (["-G", "3", "-E"], []):
        args = size + shape
        for i, s in enumerate(names):
            args += ["-i", s, "-o", str(tmp_path / ("a%d.png" % i))]
        ra = subprocess.run([exe_a] + args, capture_output=True, timeout=600)
        assert ra.returncode == 0 and b"Sanitizer" not in ra.stderr and b"runtime error" not in ra.stderr, ra.stderr.decode()[-4000:]
        assert all(open(str(tmp_path / ("a%d.png" % i)), "rb").read() == single[s] for i, s in enumerate(names))
    r = subprocess.run([exe] + small + ["-G", "2", "-i", names[1], "-o", str(tmp_path / "x.png"), "-i", str(bad), "-o", str(tmp_path / "y.png"), "-i", names[0], "-o", str(tmp_path / "z.png")], capture_output=True, timeout=600)
    assert r.returncode == 3 and b"ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_triangle_box_overlap_is_conservative(tmp_path):
    """rrtx_grid.h lists a triangle only in the cells it reaches (triangle_touches_box): the test may answer "touches" too
    often, never too rarely - random triangles (needles, slivers, collinear) against random boxes near and far."""
    exe = str(tmp_path / "tri_box_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "tri_box_check.cpp")], check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_first_bounce_records_have_slots_of_their_own(tmp_path):
    """rrtx_device.h's first_slot() against the buffer rrtx_api.cpp allocates: every (task, sample, part) inside it, no two in one slot, a wave's 64 tasks side by side;
    the code word's fields apart (tests/first_slot_check.cpp, built here with g++: the header the kernels compile)."""
    import subprocess

    exe = str(tmp_path / "first_slot_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "first_slot_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "48 layouts ok" in r.stderr, r.stderr
