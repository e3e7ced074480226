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
