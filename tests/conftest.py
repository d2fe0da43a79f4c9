import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def simlib():
    """The product's kernel sources interpreted on CPU by tests/hipsim (test infrastructure)."""
    import ctypes

    from hairfastgan_amd import _lib

    sim_dir = os.path.join(ROOT, "tests", "hipsim")
    so = os.path.join(sim_dir, "libhairfast_sim.so")
    srcs = [os.path.join(ROOT, "hairfastgan_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "hairfastgan_amd", "csrc"))
            if f.endswith((".hip", ".h"))] + [os.path.join(sim_dir, "hipsim.cpp"), os.path.join(sim_dir, "hip", "hip_runtime.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and "HIPSIM_CXX" not in os.environ:
            pytest.skip("host clang++ not available for hipsim")
        subprocess.check_call(["bash", os.path.join(sim_dir, "build_sim.sh")], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return _lib.bind(ctypes.CDLL(so))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    gdir = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return np.load(os.path.join(gdir, name))

    return load
