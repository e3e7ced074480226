"""Shared by the example scripts: repository root on sys.path, fixture paths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def eta(data):
    return max(data.get("gap") or 0.0, data.get("pinf") or 0.0, data["dinf"])
