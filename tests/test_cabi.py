"""The C ABI: the shared library loads without a GPU, exports every entry point include/rrtx.h
declares, its structs have the sizes the Python mirror assumes, and the render entry points fail
loudly (never fall back) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

import rrt_amd
from rrt_amd import _lib
from _oracle import ROOT


def _declared_functions():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rrtx_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_entry_point_is_exported():
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), "librrtx.so does not export %s" % n
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (rrtx_[a-z_0-9]+)", out))
    assert set(names) <= exported


def test_struct_sizes_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rrtx.h"\nint main(void){printf("%zu %zu %zu %zu ", sizeof(rrtx_params), sizeof(rrtx_stats), sizeof(rrtx_scene_desc), sizeof(rrtx_devinfo));'
                   'printf("%zu %zu %zu %zu %zu ", sizeof(rrtx_camera_f32), sizeof(rrtx_material_f32), sizeof(rrtx_sphere_f32), sizeof(rrtx_moving_sphere_f32), sizeof(rrtx_triangle_f32));'
                   'printf("%zu %zu %zu %zu %zu\\n", sizeof(rrtx_camera_f64), sizeof(rrtx_material_f64), sizeof(rrtx_sphere_f64), sizeof(rrtx_moving_sphere_f64), sizeof(rrtx_triangle_f64));'
                   'printf("%zu\\n", sizeof(rrtx_group_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])  # the header is plain C
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[:4] == [C.sizeof(_lib.Params), C.sizeof(_lib.Stats), C.sizeof(_lib.SceneDesc), C.sizeof(_lib.DevInfo)]
    # reference layouts (measured on the compiled reference)
    assert got[4:9] == [96, 32, 32, 56, 40]
    assert got[9:14] == [192, 40, 40, 80, 80]
    assert got[14] == C.sizeof(_lib.GroupStats)
    import rrt_amd.render as rr

    for fp64, sizes in ((False, got[4:9]), (True, got[9:14])):
        dt = rr._table_dtypes(fp64)
        assert [dt[k].itemsize for k in ("camera", "material", "sphere", "msphere", "triangle")] == sizes


def test_version_and_error_strings():
    assert _lib.lib.rrtx_version().decode().startswith("rrtx")
    # the layout version a host checks before it hands the library its structs (advisor r03: the structs grew without one)
    declared = int(re.search(r"#define\s+RRTX_ABI_VERSION\s+(\d+)", open(_lib.HEADER_PATH).read()).group(1))
    assert _lib.lib.rrtx_abi_version() == declared == _lib.ABI_VERSION
    assert isinstance(_lib.lib.rrtx_last_error(), bytes)


def test_invalid_arguments_are_rejected_without_a_device():
    p = _lib.Params()
    p.image_width, p.image_height, p.samples_per_pixel = 1, 1, 1
    h = C.c_void_p()
    assert _lib.lib.rrtx_create(C.byref(p), C.byref(h)) == -1 and not h  # RRTX_E_INVALID
    assert b"2x2" in _lib.lib.rrtx_last_error()
    assert _lib.lib.rrtx_create(None, C.byref(h)) == -1
    assert _lib.lib.rrtx_render(None, None, None) == -1
    assert _lib.lib.rrtx_set_scene(None, None) == -1
    _lib.lib.rrtx_destroy(None)  # safe on NULL
    _lib.lib.rrtx_scene_free(None)


@pytest.mark.skipif(rrt_amd.device_count() > 0, reason="checks the no-device failure mode")
def test_no_silent_fallback_without_a_gpu():
    # the product has no CPU path: creating a context without a device is a loud device error
    with pytest.raises(rrt_amd.RrtxError) as e:
        rrt_amd.Rrt(16, 16, 1, 5)
    assert e.value.code == -2
    exe = os.path.join(ROOT, "rrt")
    r = subprocess.run([exe, "-i", os.path.join(ROOT, "scenes", "test1.txt"), "-w", "16", "-h", "16", "-s", "1"], capture_output=True)
    assert r.returncode == 99 and b"HIP error" in r.stderr and r.stdout == b""
    # several devices (-G): the same loud failure, and a bad -D ends the run while the flags are parsed (main.cpp:107-110)
    r = subprocess.run([exe, "-G", "2", "-i", os.path.join(ROOT, "scenes", "test1.txt"), "-w", "16", "-h", "16", "-s", "1"], capture_output=True)
    assert r.returncode == 99 and b"HIP error" in r.stderr and r.stdout == b""
    r = subprocess.run([exe, "-D", "3", "-i", "/nonexistent/scene.txt"], capture_output=True)
    assert r.returncode == 99  # (not 2: the device is selected before the scene is opened)
    with pytest.raises(rrt_amd.RrtxError) as e:
        rrt_amd.RrtGroup(2, 16, 16, 1, 5)
    assert e.value.code == -2


def test_product_does_not_reference_the_oracle():
    # parity claims are void if the product path can route through oracle/ (or any CPU fallback)
    for base in ("rrt_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    # (comments may cite the oracle; what must not exist is a link, include, import or call)
                    assert "librrt_oracle" not in text and "rrto_" not in text, os.path.join(dirpath, f)
                    assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', text), os.path.join(dirpath, f)
                    assert not re.search(r"^\s*(from|import)\s+\S*oracle", text, flags=re.M), os.path.join(dirpath, f)
    libs = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in libs and "libamdhip64" in libs


def test_reference_side_binding_is_compiled_against_the_reference_headers():
    """oracle/_ref/rrt_dropin = the reference's own main.cpp + oracle/ref_dropin.cpp (class Rrt of rrt.h:14-48 over
    librrtx.so), built by oracle/Makefile where /root/reference exists: INTEGRATION.md section 1 as a binary.  Here
    (no GPU): it exists, binds the C ABI's entry points dynamically, and fails like check_cuda without a device;
    what it renders is compared with `rrt` on the GPU (tests/test_gpu_dropin.py)."""
    from _oracle import REF_DIR, have_reference

    if not have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    for exe in ("rrt_dropin", "rrtd_dropin"):
        path = os.path.join(REF_DIR, exe)
        assert os.path.exists(path), path
        undefined = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        for sym in ("rrtx_create", "rrtx_set_scene", "rrtx_render", "rrtx_destroy", "rrtx_last_error"):
            assert re.search(r"\bU %s\b" % sym, undefined), (exe, sym)
        assert "librrtx.so" in subprocess.run(["ldd", path], capture_output=True, text=True).stdout
        r = subprocess.run([path], capture_output=True)
        assert r.returncode == 1 and b"ERROR: no scene loaded." in r.stderr  # the reference's main.cpp:126-128
        if rrt_amd.device_count() == 0:
            r = subprocess.run([path, "-i", os.path.join(ROOT, "scenes", "test1.txt"), "-w", "16", "-h", "16", "-s", "1"], capture_output=True)
            assert r.returncode == 99 and b"HIP error" in r.stderr and r.stdout == b""


def test_integration_md_shows_the_file_that_is_compiled():
    # the binding INTEGRATION.md section 1 tells a maintainer to add IS oracle/ref_dropin.cpp (compiled and tested), not a sketch
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "oracle", "ref_dropin.cpp")).read()
    body = src[src.index('#include "rrt.h"'):]
    assert body in doc


def test_bench_profiles_the_kernels_that_exist():
    """bench.py finds its two kernels in rocprofv3's output by name (LIST_KERNEL, ACCEL_KERNEL): each must match exactly one
    render kernel among the library's symbols per scene class - a template parameter added to render_kernel must not
    silently turn the roofline into its fallback."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_names", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = subprocess.run(["nm", "-C", "--defined-only", os.path.join(ROOT, "rrt_amd", "librrtx.so")], capture_output=True, text=True, check=True).stdout
    kernels = sorted({line.split(" ", 2)[2].split("(")[0].replace("void ", "") for line in out.splitlines() if " rrtx::render_kernel<" in line and "__device_stub__" not in line})
    assert len(kernels) >= 20
    lists = [k for k in kernels if bench.LIST_KERNEL in k]
    accel = [k for k in kernels if bench.ACCEL_KERNEL in k]
    # behind the prefix stands SOV, the variant a launch selects by what it knows (rrtx_kernels.hip): the matrix-core list scan general / plain; the walk kernels
    # for scenes of every kind, of spheres alone, with a first-bounce pre-pass, and the plain forms of the latter two - one of them runs per launch
    assert sorted(k.rsplit(", ", 1)[1] for k in lists) == ["0>", "4>"], lists
    assert sorted(k.rsplit(", ", 1)[1] for k in accel) == ["0>", "1>", "2>", "5>", "6>"], accel
    for name in (bench.MESH_RENDER_KERNEL, bench.MESH_RESUME_KERNEL):  # the mesh sub-result's two passes: exactly one kernel each
        assert len([k for k in kernels if name in k]) == 1, name
