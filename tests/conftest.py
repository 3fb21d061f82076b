import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Several ranks of a sharded frame in ONE process on ONE device (tests/test_p2p_gpu.py, world up to 8) wait for each other inside
# their kernels: each rank's stream needs a hardware queue of its own (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 queues by default).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import fastlivo  # noqa: E402,F401  registers the package fast_livo_amd


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def scene():
    from fast_livo_amd import synth
    return synth.make_scene()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def emul_lib():
    """Host build of the product's per-measurement arithmetic (tests/host_emul) -- test-only."""
    import ctypes
    d = os.path.join(ROOT, "tests", "host_emul")
    so = os.path.join(d, "libemul.so")
    src = os.path.join(d, "emul.cpp")
    hdr = os.path.join(ROOT, "fast-livo_amd", "csrc", "fl_math.h")
    hdrs = [hdr, os.path.join(ROOT, "fast-livo_amd", "csrc", "fl_ikfom_math.h"), os.path.join(ROOT, "fast-livo_amd", "csrc", "exact_chain.h")]
    hdrs = [h for h in hdrs if os.path.exists(h)]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def emul_series_lib():
    """The same host build with -DFL_IK_SERIES_ON_HOST: the Mode-23 manifold operations in the power-series forms the DEVICE compiles
    (fl_ikfom_math.h FL_IK_SERIES) -- what ships, run on the CPU."""
    import ctypes
    d = os.path.join(ROOT, "tests", "host_emul")
    so = os.path.join(d, "libemul_series.so")
    src = os.path.join(d, "emul.cpp")
    hdrs = [os.path.join(ROOT, "fast-livo_amd", "csrc", h) for h in ("fl_math.h", "fl_ikfom_math.h", "exact_chain.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-DFL_IK_SERIES_ON_HOST", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def gpu_lib():
    if not _has_gpu():
        pytest.skip("no GPU in this container")
    from fast_livo_amd import capi
    return capi
