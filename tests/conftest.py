import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Both schedules of a small synchronous frame (launch chain, fused bounce loop) leave the same bits, and a context TIMES them against each other over
# its first frames (pt_stats.schedule).  The suite pins which one runs where it matters (PT_FUSED, the `sched` fixture of test_gpu_parity.py), so by
# default the measurement is off here: which schedule a frame took — and with it the launch counts some tests assert — must not depend on timing.
# tests/test_gpu_schedule.py::test_online_schedule_choice_* turn it on.
os.environ.setdefault("PT_SCHED_TRIALS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc_det():
    from oracle import orc

    return orc.Oracle("det")


@pytest.fixture(scope="session")
def orc_libm():
    from oracle import orc

    return orc.Oracle("libm")


@pytest.fixture(scope="session")
def ptlib():
    """The product library; GPU tests fail (not skip) if it is missing — there is no fallback path."""
    from optixpathtracer_amd import _lib

    return _lib.load_library()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    neq = bits(a) != bits(b)
    # +0 / -0 are the same value for every consumer of these buffers
    neq &= ~((a == 0) & (b == 0))
    n = int(neq.sum())
    if n:
        idx = np.argwhere(neq)[:5]
        ex = [(tuple(i), float(a[tuple(i)]), float(b[tuple(i)])) for i in idx]
        raise AssertionError(f"{what}: {n}/{a.size} elements differ bitwise, e.g. {ex}")
