"""SharedDiv (the window kernels' division, kernels_windows_exact.hip.inc) is the CORRECTLY ROUNDED quotient:
CPU replay of its exact fma sequence for every tabulated denominator against `/`.
The other half of the argument - every tabulated reciprocal equals the IEEE quotient 1/den, and the in-kernel
constructor does for every denominator up to 2^19 - is checked by plaac_ctx_create on the device (a context
cannot be created otherwise; test_gpu_parity.py::test_reciprocal_selftest_passes_at_context_creation)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def replay(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("sdiv") / "libshared_div_replay.so")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-march=native", "-shared", "-fPIC", "-o", so,
                    os.path.join(HERE, "shared_div_replay.c"), "-lm"], check=True)
    L = C.CDLL(so)
    L.shared_div_replay.argtypes = [C.c_long, C.POINTER(C.c_double)]
    L.shared_div_replay.restype = C.c_long
    return L


def test_markstein_step_with_exact_reciprocal_is_the_ieee_quotient(replay):
    out = (C.c_double * 4)()
    bad = replay.shared_div_replay(1500, out)
    assert int(out[1]) == 41 + 41 * 41
    assert out[0] > 3e7  # cases
    assert bad == 0, "first mismatch: a=%r d=%r" % (out[2], out[3])


def test_hard_cases_do_break_a_plain_multiply():
    """the hard-case generator is not vacuous: a * RN(1/d) alone is NOT the quotient for many of them"""
    rng = np.random.default_rng(0)
    d = 7.0
    y = 1.0 / d
    a = rng.random(100000) * 100
    assert np.count_nonzero(a * y != a / d) > 1000
