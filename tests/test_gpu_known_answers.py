"""GPU solves (all through the C ABI) against every optimal value the reference ships for the three entry points of this
path: data/sdplib/README:39-51 (gpp -> ManiSDP_unitdiag), :71-88 (maxG*, mcp* -> ManiSDP_onlyunitdiag), :98-105 (theta ->
ManiSDP_unittrace).  The README prints 5-7 significant digits; a value must agree to exactly those digits (conftest.within_print)
-- that is the accuracy of the pin, 1e-7 relative for the 7-digit families.  The oracle passes the same assertions on the CPU
(tests/test_oracle_known_answers.py)."""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bqp_bruteforce_min, golden_path, within_print

pytestmark = pytest.mark.gpu

PRINTED = json.load(open(golden_path("known_answers_printed.json")))
THETA_OPTS = dict(tol=1e-8, TR_maxiter=30, TR_maxinner=200)       # tests/test_oracle_known_answers.py: the budget under which the family converges
GPP_OPTS = dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6)


def _sdpa(name):
    from manisdp_matlab_amd import problems
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float).ravel()
    return At, b, c, K


@pytest.mark.parametrize("name", ["mcp100", "mcp124-1", "mcp124-2", "mcp124-3", "mcp124-4", "mcp250-1", "mcp250-2", "mcp250-3",
                                  "mcp250-4", "mcp500-1", "mcp500-2", "mcp500-3", "mcp500-4"])
@pytest.mark.parametrize("eig", ["host", "device"])
def test_onlyunitdiag_mcp(name, eig):
    from manisdp_matlab_amd import solvers
    At, b, c, K = _sdpa(name)
    n = K["s"]
    C = sp.csr_matrix(c.reshape(n, n, order="F"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": eig}, verbose=False)
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert within_print(-obj, PRINTED[name])
    assert np.abs(np.linalg.norm(Y, axis=1) - 1.0).max() < 1e-12


@pytest.mark.parametrize("eig", ["host", "device"])
def test_onlyunitdiag_maxG32(eig):
    """Gset G32 (n = 2000): the device escape takes the block eigen-solver at this size (n >= 2048 is its default threshold;
    forced here for n = 2000 so that a README value pins that path too)."""
    from manisdp_matlab_amd import problems, solvers, _lib
    C = problems.maxcut_cost_matrix(golden_path("G32.txt.gz"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": eig, "escape_method": 2}, verbose=False)
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert within_print(-obj, PRINTED["maxG32"])
    if eig == "device":
        assert data.get("escape_method") == 1


def test_onlyunitdiag_maxG60():
    """README:75, Gset G60, n = 7000 (7 digits), device escape (the block eigen-solver at this size).  (maxG51 and maxG55 pin
    nothing: the values README:73-74 prints belong to neither the Gset graphs nor the .dat-s files of the tree, see
    tests/golden/make_fixtures.py.)"""
    from manisdp_matlab_amd import problems, solvers
    C = problems.maxcut_cost_matrix(golden_path("G60.txt.gz"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": "device"}, verbose=False)
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert within_print(-obj, PRINTED["maxG60"])
    assert data.get("escape_method") == 1


@pytest.mark.parametrize("name", ["gpp100", "gpp124-1", "gpp124-2", "gpp124-3", "gpp124-4", "gpp250-1", "gpp250-2", "gpp250-3", "gpp250-4",
                                  "gpp500-1", "gpp500-2", "gpp500-3", "gpp500-4"])
def test_unitdiag_gpp(name):
    from manisdp_matlab_amd import solvers
    At, b, c, K = _sdpa(name)
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, dict(GPP_OPTS), verbose=False)
    # the family crawls at eta = 5e-8 ... 1e-6 for hundreds of outer iterations (no strictly feasible point: <J, X> = 0 with X >= 0;
    # DESIGN.md section 5) and where exactly a run stops moves with the summation order of the kernels (gpp500-1: 6e-7 with the
    # separate SDDMM launches of round 3, 1.06e-6 with the fused one) -- the printed digits are the pin
    assert max(data["gap"], data["pinf"], data["dinf"]) < 2e-6
    assert within_print(-obj, PRINTED[name])


@pytest.mark.parametrize("name", ["theta1", "theta2", "theta3", "theta4", "theta5"])
@pytest.mark.parametrize("eig", ["host", "device"])
def test_unittrace_theta(name, eig):
    from manisdp_matlab_amd import solvers
    At, b, c, K = _sdpa(name)
    Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, dict(THETA_OPTS, eig=eig), verbose=False)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert within_print(-obj, PRINTED[name])
    assert abs(-obj - float(PRINTED[name])) < 1e-7 * float(PRINTED[name])
    assert abs(np.linalg.norm(Y) - 1.0) < 1e-12


def test_unittrace_theta6_as_far_as_it_gets():
    """theta6 (README:103, n = 300, m = 4375) is the one instance of the family that does NOT reach KKT 1e-8 under the option set
    above: the reference's algorithm leaves through "Slow progress" at residues of 4e-5 after 60 outer iterations.  The value it
    stops at is pinned to what that residual allows -- 3e-7 relative, two units of the README's last digit."""
    from manisdp_matlab_amd import solvers
    At, b, c, K = _sdpa("theta6")
    Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, dict(THETA_OPTS, eig="host"), verbose=False)
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-3
    assert abs(-obj - float(PRINTED["theta6"])) < 5e-7 * float(PRINTED["theta6"])



@pytest.mark.parametrize("d", [10, 20])
@pytest.mark.parametrize("eig", ["host", "device"])
def test_bqp_relaxation_is_bounded_by_the_bruteforce_minimum(d, eig):
    """Generator + ManiSDP_unitdiag pinned WITHOUT the oracle (src/basicfunction/bqpmom.m:1-5, example/example_bqp.m:36-43):
    the relaxation value is a lower bound of min over {-1,1}^d of x'Qx + e'x (2^d points enumerated on the host), with
    equality when the returned X has rank one -- then its first column is a sign vector that attains the value."""
    from manisdp_matlab_amd import problems, solvers
    Q = np.loadtxt(golden_path(f"bqp_Q_{d}_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path(f"bqp_e_{d}_1.txt.gz"), delimiter=",")
    fmin, xmin = bqp_bruteforce_min(Q, e)
    At, b, c, K = problems.bqpmom(d, Q, e)
    c = np.asarray(c.todense()).ravel()
    maxc = np.abs(c).max()
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c / maxc, K, {"tol": 1e-8, "eig": eig}, verbose=False)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    val = obj * maxc
    assert val <= fmin + 1e-6 * max(1.0, abs(fmin))
    sv = np.linalg.svd(Y, compute_uv=False)
    if sv[1] < 1e-6 * sv[0]:
        assert abs(val - fmin) <= 1e-6 * max(1.0, abs(fmin))
        X = Y @ Y.T
        x = np.sign(X[1:d + 1, 0])
        assert abs(x @ Q @ x + e @ x - fmin) <= 1e-9 * max(1.0, abs(fmin))


@pytest.mark.parametrize("eig", ["host", "device"])
def test_unitdiag_thetaG11(eig):
    """data/sdplib/README:104: thetaG11 (n = 801, m = 2401) is a unit-diagonal instance (tests/test_oracle_known_answers.py checks
    the structure) -> ManiSDP_unitdiag, default options, 7 printed digits."""
    from manisdp_matlab_amd import solvers
    At, b, c, K = _sdpa("thetaG11")
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {"tol": 1e-8, "eig": eig}, verbose=False)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert within_print(-obj, PRINTED["thetaG11"])
    assert np.abs(np.linalg.norm(Y, axis=1) - 1.0).max() < 1e-12


def test_unitdiag_thetaG51():
    """data/sdplib/README:105: thetaG51 (n = 1001, m = 6910, printed 3.49000e+02), a unit-diagonal instance like thetaG11.  With the
    reference's defaults the algorithm stops at 349.0074 (status 1; the oracle likewise, 300 s on the CPU); with the option set of
    the gpp family (tests above) the value reaches the six printed digits within 60 outer iterations while the residues crawl
    (pinf 2e-5 at iteration 60, 4e-6 at 120: tools/archive/thetaG51_opts.py) -- as for gpp, the digits are the pin."""
    from manisdp_matlab_amd import solvers
    At, b, c, K = _sdpa("thetaG51")
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, dict(GPP_OPTS, tol=1e-6, AL_maxiter=60, eig="host"), verbose=False)
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-4
    assert within_print(-obj, PRINTED["thetaG51"])


@pytest.mark.parametrize("k,d,theta", [(5, [1], 16.0), (7, [5, 6], 128.0 / 3.0), (8, [1], 128.0), (6, [6], 32.0)])
def test_unittrace_hamming_graphs(k, d, theta):
    """problems.generate_hamming (example/generate_hamming.m:24-59) through ManiSDP_unittrace on the GPU: hypercubes and the antipodal
    matching (perfect graphs: theta = 2^(k-1)), and H_{7,{5,6}} -- SDPLIB's hamming_7_5_6, whose value is 128/3."""
    from manisdp_matlab_amd import problems, solvers
    At, b, c, K = problems.generate_hamming(k, d)
    Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, dict(THETA_OPTS, eig="host"), verbose=False)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert abs(-obj - theta) <= 1e-7 * theta


@pytest.mark.parametrize("N,seed", [(10, 1), (50, 1)])
def test_unittrace_rotation_search(N, seed):
    """example/example_rotationsearch.m (N = 50, half outliers: n = 204, m = 8151) on the GPU: tight relaxation, inliers and
    rotation recovered (tests/test_oracle_known_answers.py::check_rotation_search); N = 10 also against the oracle's optimum."""
    from manisdp_matlab_amd import solvers
    from test_oracle_known_answers import check_rotation_search

    def solve(At, b, c, K, o):
        return solvers.ManiSDP_unittrace(At, b, c, K, dict(o, eig="host"), verbose=False)

    fval = check_rotation_search(solve, N, seed)
    if N == 10:
        from oracle import manisdp_ref as R
        assert abs(fval - check_rotation_search(R.ManiSDP_unittrace, N, seed)) <= 1e-7 * abs(fval)

