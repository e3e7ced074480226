"""CPU tests (no GPU): pin the oracle against every known answer the reference ships for this
path (data/sdplib/README:39-51,71-88,98-105 -> tests/golden/known_answers.json) and against the
SURVEY.md probe value for G1.  These are what make the oracle trustworthy as the GPU checker."""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_path
from manisdp_matlab_amd import problems
from oracle import manisdp_ref as R

KNOWN = json.load(open(golden_path("known_answers.json")))


def _mcp(name):
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    n = K["s"]
    return sp.csr_matrix(c.toarray().reshape(n, n, order="F"))


@pytest.mark.parametrize("name", ["mcp100", "mcp124-1", "mcp250-1"])
def test_onlyunitdiag_sdplib_mcp(name):
    """max <F0,X>, X_ii = 1  ->  ManiSDP_onlyunitdiag with C = -F0 returns -value (7 printed digits)."""
    Y, obj, data = R.ManiSDP_onlyunitdiag(_mcp(name), {})
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert abs(-obj - KNOWN[name]) < 1e-6 * abs(KNOWN[name])
    assert np.allclose(np.linalg.norm(Y, axis=1), 1.0, atol=1e-12)


def test_onlyunitdiag_gset():
    """maxG11 == Gset G11 with C = -L/4 (example_maxcut.m:10-11); G1 against the survey probe."""
    Y, obj, data = R.ManiSDP_onlyunitdiag(problems.maxcut_cost_matrix(golden_path("G11.txt.gz")), {})
    assert data["dinf"] < 1e-8
    assert abs(-obj - KNOWN["maxG11"]) < 1e-6 * KNOWN["maxG11"]
    Y, obj, data = R.ManiSDP_onlyunitdiag(problems.maxcut_cost_matrix(golden_path("G1.txt.gz")), {})
    assert data["dinf"] < 1e-8
    assert abs(obj - (-12083.19765455)) < 1e-6 * 12083.2


def test_quirk_q1_variants_agree_at_optimum():
    """Reference behaviour (stale eG after a rejected step) and per-point state reach the same optimum."""
    C = problems.maxcut_cost_matrix(golden_path("G11.txt.gz"))
    rng = np.random.default_rng(1)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    _, o1, d1 = R.ManiSDP_onlyunitdiag(C, {"Y0": Y0}, q1="reference")
    _, o2, d2 = R.ManiSDP_onlyunitdiag(C, {"Y0": Y0}, q1="correct")
    assert d1["dinf"] < 1e-8 and d2["dinf"] < 1e-8
    assert abs(o1 - o2) < 1e-7 * abs(o1)


def test_unitdiag_gpp100():
    """gpp100 through fromsdpa: constraint 1 is <J,X> = 0, the rest X_ii = 1; README gives 6 digits.
    Option set of SURVEY.md section 4 (the defaults are tuned for BQP)."""
    At, b, c, K = problems.from_sdpa(golden_path("gpp100.dat-s.gz"))
    opts = dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6)
    Y, obj, data = R.ManiSDP_unitdiag(At, b, c, K, opts)
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-6
    assert abs(-obj - KNOWN["gpp100"]) < 2e-5 * abs(KNOWN["gpp100"])


def test_unitdiag_bqp_kkt_self_certification():
    """BQP instances have no stored optimum: KKT residues below tol certify the value."""
    Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(10, Q, e)
    c = np.asarray(c.todense()).ravel()
    Y, obj, data = R.ManiSDP_unitdiag(At, b, c / np.abs(c).max(), K, {})
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    X = Y @ Y.T
    assert np.allclose(np.diag(X), 1.0, atol=1e-12)
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) / (1 + np.linalg.norm(b)) < 1e-8


def test_unittrace_theta1():
    """theta1: constraint 1 is tr X = 1, F0 = J; options of example/example_theta.m:50-53."""
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    Y, obj, data = R.ManiSDP_unittrace(At, b, c, K, dict(tol=1e-6, sigma0=1e5, sigma_max=1e8))
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-5
    assert abs(-obj - KNOWN["theta1"]) < 1e-5 * KNOWN["theta1"]
    assert abs(np.linalg.norm(Y) - 1.0) < 1e-12


@pytest.mark.parametrize("name", ["mcp100", "mcp124-1"])
def test_generic_sdplib_mcp(name):
    """The generic entry point (src/primal/ManiSDP.m, Euclidean manifold) on the same SDPLIB instances, treated
    as plain A(X) = b problems: same README values."""
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    Y, obj, data = R.ManiSDP(At, b, c, K, {})
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert abs(-obj - KNOWN[name]) < 1e-6 * abs(KNOWN[name])


def test_generic_quartic_on_sphere_kkt_self_certification():
    """example/example_qsphere.m:18-27 with the reference's coefficient file data/qs_c_10_1.txt: the reference stores
    no optimum for these instances, so the result is pinned by the KKT residues (gap, pinf, dinf < 1e-8 certify the
    objective to ~1e-8 by weak duality) and by the constraints A(X) = b of the moment matrix X = Y Y'."""
    import numpy as np
    coe = np.loadtxt(golden_path("qs_c_10_1.txt.gz"), delimiter=",").ravel()
    At, b, c, K = problems.qsmom(10, coe)
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    Y, obj, data = R.ManiSDP(At, b, c, K, {})
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    X = Y @ Y.T
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) / (1 + np.linalg.norm(b)) < 1e-8
    assert abs(obj - c @ X.ravel(order="F")) < 1e-9 * max(1.0, abs(obj))


def test_multiblock_direct_sum_of_two_maxcut_problems():
    """ManiSDP_multiblock.m restated (oracle): mcp100 (+) mcp124-1 as one two-block SDP with both unit diagonals in the
    product manifold has the sum of the two SDPLIB optima (data/sdplib/README:76-77)."""
    import scipy.sparse as sp
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    known = json.load(open(golden_path("known_answers.json")))
    cs, ns = [], []
    for name in ("mcp100", "mcp124-1"):
        At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
        cs.append(np.asarray(c.todense()).ravel()); ns.append(K["s"])
    c = np.concatenate(cs)
    At = sp.csc_matrix((np.ones(1), ([0], [0])), shape=(c.size, 1))
    Y, obj, d = R.ManiSDP_multiblock(At, np.ones(1), c, {"s": ns, "nob": 2}, {})
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    want = -(known["mcp100"] + known["mcp124-1"])
    assert abs(obj - want) <= 1e-6 * abs(want)


def test_dual_oracle_agrees_with_primal_oracle():
    """ManiDSDP_unitdiag on the SOS relaxation of a BQP (bqpsos.m data as example_bqp_dual.m builds it) and
    ManiSDP_unitdiag on the moment relaxation of the same BQP (bqpmom.m) are a primal/dual pair: same optimum.  Pins the
    dual restatement to the primal one, which is pinned to the SDPLIB values above."""
    import numpy as np
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    for d, seed in ((6, 0), (9, 4)):
        rng = np.random.default_rng(seed)
        Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
        e = rng.standard_normal(d)
        At, b, c, K = problems.bqpmom(d, Q, e)
        _, fp, dp = R.ManiSDP_unitdiag(At, b, c, K, {"tol": 1e-8}, rng=np.random.default_rng(0))
        A, bs, cs, Ks, dAAt, maxb = problems.bqpsos_dual_problem(Q, e, d)
        assert Ks["s"] == K["s"] and np.allclose(dAAt, np.asarray(A[:, 1:].multiply(A[:, 1:]).sum(axis=1)).ravel())
        for ls in (0, 1):
            _, fd, dd = R.ManiDSDP_unitdiag(A, bs, cs, Ks, {"tol": 1e-8, "dAAt": dAAt, "line_search": ls}, rng=np.random.default_rng(0))
            assert dd["status"] == 0 and max(dd["gap"], dd["pinf"], dd["dinf"]) < 1e-8
            assert abs(fd * maxb - fp) <= 1e-6 * max(1.0, abs(fp))
