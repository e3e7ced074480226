"""CPU test: the C-ABI library loads and exports every symbol include/manisdp_hip.h declares, and the
ctypes table binds exactly that set (no compute calls are made: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "manisdp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msdp_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ["msdp_create_onlyunitdiag_csc", "msdp_create_onlyunitdiag_dense", "msdp_create_affine", "msdp_rtr",
                 "msdp_hessvec", "msdp_cost", "msdp_rgrad", "msdp_proj", "msdp_retr", "msdp_set_multipliers",
                 "msdp_escape_eigs", "msdp_comm_init", "msdp_destroy", "msdp_last_error"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from manisdp_matlab_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table covers the header one-to-one
    assert sorted(_lib.SIGNATURES) == _declared()
    assert lib.msdp_version is not None


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product path must fail, not fall back."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    C = problems.toroidal_grid_maxcut(4, 4)
    with pytest.raises(_lib.MsdpError):
        _lib.Handle.onlyunitdiag(C)


def test_caller_memory_never_reaches_the_runtime_copy_calls():
    """Every host <-> device copy of the library goes through msdp_xfer.hip (pinned staging of unpinned caller memory): no other
    source file calls the hipMemcpy family (round 6: the runtime registers pageable ranges it copies from, and the process stalls for
    20 - 35 ms when their owner frees them -- msdp_xfer.hip's header has the measurements)."""
    import glob
    csrc = os.path.join(ROOT, "manisdp-matlab_amd", "csrc")
    bad = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        if os.path.basename(path) == "msdp_xfer.hip":
            continue
        txt = re.sub(r"//[^\n]*", "", open(path).read())
        for m in re.finditer(r"\bhipMemcpy\w*\s*\(", txt):
            bad.append((os.path.basename(path), m.group(0)))
    assert not bad, bad
