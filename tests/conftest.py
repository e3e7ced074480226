import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def golden_path(name):
    return os.path.join(GOLDEN, name)


def printed_ulp(printed):
    """One unit of the last digit of a value as the reference prints it ("-4.49435e+01" -> 1e-4)."""
    mant, exp = printed.lower().split("e")
    digits = len(mant.replace("-", "").replace("+", "").replace(".", ""))
    return 10.0 ** (int(exp) - (digits - 1))


def within_print(value, printed, slack=0.6):
    """value agrees with a README value to the digits the README prints: |value - printed| <= slack units of the last
    printed digit (0.5 = exact rounding; the default leaves a 20 % cushion for values that sit next to a rounding boundary)."""
    return abs(value - float(printed)) <= slack * printed_ulp(printed)


def bqp_bruteforce_min(Q, e):
    """min over x in {-1,1}^d of x'Qx + e'x by enumeration (d <= 22): the quantity the second-order moment relaxation of
    src/basicfunction/bqpmom.m:1-5 bounds from below -- a pin of generator + solver that does not pass through the oracle."""
    import numpy as np
    d = len(e)
    assert d <= 22
    best, arg = np.inf, None
    lo_bits = min(d, 12)
    codes = np.arange(1 << lo_bits)
    Xlo = 1.0 - 2.0 * ((codes[:, None] >> np.arange(lo_bits)[None, :]) & 1)           # all sign patterns of the first lo_bits variables
    for hi in range(1 << (d - lo_bits)):
        xhi = 1.0 - 2.0 * ((hi >> np.arange(d - lo_bits)) & 1)
        X = np.hstack([Xlo, np.broadcast_to(xhi, (Xlo.shape[0], d - lo_bits))])
        f = np.einsum("ij,jk,ik->i", X, Q, X) + X @ e
        k = int(np.argmin(f))
        if f[k] < best:
            best, arg = float(f[k]), X[k].copy()
    return best, arg
