"""Host <-> device copies of caller memory (msdp_xfer.hip, round 6): every copy goes through a pinned staging buffer of two 16-MB halves.
The cases here cross the half: several chunks per copy, 1-D (the factor: set_point / get_point) and 2-D (a dense C whose rows go to a
padded leading dimension), and the run-time switch that hands copies to the runtime as before must give the same bits."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_factor_round_trip_across_several_staging_chunks():
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    rows, cols, p = 400, 800, 20                       # n = 320 000 rows x 20 columns = 51 MB: four chunks of 16 MB each way
    C = problems.toroidal_grid_maxcut(rows, cols, seed=5)
    n = C.shape[0]
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((n, p))
    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    Yg = h.get_point()
    assert np.array_equal(Yg, Y)                       # copies, a pack kernel and its inverse: every bit comes back
    z = h.get_z()                                      # z = sum((Y*C).*Y) (ManiSDP_onlyunitdiag.m:46-47) against the host's
    zref = np.sum((C @ Y) * Y, axis=1)
    assert np.abs(z - zref).max() <= 1e-12 * max(1.0, np.abs(zref).max())
    h.close()


def test_dense_cost_matrix_goes_up_in_strided_chunks():
    from manisdp_matlab_amd import _lib
    _lib.load()
    n, p = 2100, 8                                     # 2100 x 2100 doubles = 35 MB: three 2-D chunks (host pitch n, device pitch roundup(n, 16))
    rng = np.random.default_rng(2)
    A = rng.standard_normal((n, n))
    Cd = 0.5 * (A + A.T)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h = _lib.Handle.onlyunitdiag(Cd, pcap=p)
    h.set_point(Y)
    from oracle import manisdp_ref as R
    U -= Y * np.sum(Y * U, axis=1, keepdims=True)
    H = h.hessvec(U)                                   # ManiSDP_onlyunitdiag.m:127-130, against the oracle's restatement
    ref = R.hessvec_onlyunitdiag(Cd, Y, U)
    assert np.linalg.norm(H - ref) <= 1e-12 * np.linalg.norm(ref)
    assert abs(h.cost() - R._OnlyUnitDiagProblem(Cd, n, p).cost(Y)) <= 1e-12 * abs(h.cost())
    h.close()


def test_staged_copies_and_runtime_copies_give_the_same_bits(tmp_path):
    """MSDP_XFER_DIRECT_MAX (read once per process): the same small solve in two fresh processes."""
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from manisdp_matlab_amd import _lib, problems\n"
            "C = problems.toroidal_grid_maxcut(40, 50, seed=3); n = C.shape[0]\n"
            "Y = np.random.default_rng(0).standard_normal((n, 12)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)\n"
            "h = _lib.Handle.onlyunitdiag(C, pcap=12); h.set_point(Y)\n"
            "st = h.rtr(_lib.default_opts(maxiter=5, maxinner=30, tolgradnorm=1e-9))\n"
            "np.save(sys.argv[1], h.get_point()); h.close()\n" % ROOT)
    outs = []
    for tag, val in (("staged", "0"), ("direct", "100000000000")):
        out = str(tmp_path / (tag + ".npy"))
        env = dict(os.environ, MSDP_XFER_DIRECT_MAX=val)
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])
