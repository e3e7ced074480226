"""CPU test: every Python script of the repository parses (examples, tools, diagnostics, bench, entry points) -- a syntax
error in a script that only runs on the GPU box would otherwise surface there."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "examples", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "archive", "*.py")) +
                 glob.glob(os.path.join(ROOT, "tests", "diagnostics", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "golden", "*.py")) +
                 [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_parses(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)


def test_examples_cover_the_self_contained_reference_examples():
    """example/*.m of the reference that need nothing outside its tree (no spotless / STRIDE): each has a counterpart."""
    have = {os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "examples", "example_*.py"))}
    for name in ("example_maxcut.py", "example_bqp.py", "example_bqp_dual.py", "example_theta.py", "example_qsphere.py",
                 "example_matrixcompletion.py", "example_bqp_sparse.py", "example_qsphere_sparse.py"):
        assert name in have
