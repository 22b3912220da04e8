import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side artefacts every test tier needs: the oracle, and the product library (hipcc
    # cross-compiles it without a GPU).  Both are no-ops when already built.
    lib = os.path.join(ROOT, "rrt_amd", "librrtx.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(ROOT, "rrt")):
        subprocess.check_call(["make", "-C", ROOT], stdout=subprocess.DEVNULL)
    from _oracle import build_oracle

    build_oracle()


def _gpu_present():
    try:
        import rrt_amd

        return rrt_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail (not skip) when the HIP path cannot run: a silent skip would hide a broken build."""
    import rrt_amd

    assert rrt_amd.device_count() > 0, "no HIP device visible: -m gpu tests must run on the MI355X box"
    return rrt_amd
