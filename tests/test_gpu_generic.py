"""GPU parity tests for the generic entry point ManiSDP(At, b, c, K, options) (reference src/primal/ManiSDP.m,
Euclidean manifold): operators through the C-ABI against the oracle closures (1e-11 relative),
single-tCG agreement, and the SDPLIB known answer the reference ships for mcp100 (README table)."""
import json

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


def _sdpa(name):
    from manisdp_matlab_amd import problems
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    return At, np.asarray(b, float), np.asarray(c.todense()).ravel(), K


@pytest.mark.parametrize("case,p", [("mcp100", 1), ("gpp100", 6), ("theta1", 20), ("mcp124-1", 33), ("theta2", 140)])
def test_generic_operators(lib, case, p):
    from oracle import manisdp_ref as R
    At, b, c, K = _sdpa(case)
    n = K["s"]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p))
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    sigma = 0.73
    prob = R._GenericProblem(At, b, c, n, p)
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_GENERIC, At, b, c, n, pcap=p)
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    assert _relerr(h.proj(U), U) == 0.0                        # euclideanfactory.m:59
    assert _relerr(h.retr(U), Y + U) < 1e-15                   # euclideanfactory.m:67-76
    Z = Y + 0.5 * U
    assert abs(h.linesearch_cost(U, 0.5) - prob.cost(Z)) <= 1e-11 * max(1.0, abs(prob.cost(Z)))
    assert _relerr(h.get_point(), Y) == 0.0
    h.close()


def test_generic_rtr_single_tcg(lib):
    """maxiter = 1: one tCG; Hess-vec count, stop code and cost agree with the oracle (Delta0 from
    typicaldist = sqrt(n*p), euclideanfactory.m:57)."""
    from oracle import manisdp_ref as R, manopt_rtr
    At, b, c, K = _sdpa("gpp100")
    n, p = K["s"], 5
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p))
    y = np.zeros(b.size); sigma = 1e-2
    h = lib.Handle.affine(lib.KIND_GENERIC, At, b, c, n)
    h.set_multipliers(y, sigma)
    for maxinner in (1, 5, 20):
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
        prob = R._GenericProblem(At, b, c, n, p)
        prob.y, prob.sigma = y, sigma
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        assert st.hessvecs == info.hessvecs
        assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-10 * max(1.0, abs(f_ref))
    h.close()


def test_generic_solver_mcp100_known_answer(lib):
    """mcp100 through the generic solver (defaults of ManiSDP.m:9-25): SDPLIB value 226.1574, KKT residues < 1e-8."""
    from manisdp_matlab_amd import solvers
    known = json.load(open(golden_path("known_answers.json")))
    At, b, c, K = _sdpa("mcp100")
    Y, obj, data = solvers.ManiSDP(At, b, c, K, {}, verbose=False, rng=np.random.default_rng(0))
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert abs(-obj - known["mcp100"]) < 1e-6 * known["mcp100"]
    X = Y @ Y.T
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) / (1 + np.linalg.norm(b)) < 1e-8


def _qsphere(d):
    """Quartic on the sphere, second-order moment relaxation (src/basicfunction/qsmom.m; the reference solves it with
    the generic ManiSDP, example/example_qsphere.m:18-27).  d = 10 uses the reference's coefficient file."""
    from manisdp_matlab_amd import problems as P
    if d == 10:
        coe = np.loadtxt(golden_path("qs_c_10_1.txt.gz"), delimiter=",").ravel()
    else:
        coe = np.random.default_rng(5).standard_normal(P.get_basis(d, 4).shape[1])
    At, b, c, K = P.qsmom(d, coe)
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    return At, b, c, K


@pytest.mark.parametrize("d,p", [(10, 4), (10, 40), (16, 21)])
def test_generic_operators_quartic_on_sphere(lib, d, p):
    from oracle import manisdp_ref as R
    At, b, c, K = _qsphere(d)
    n = K["s"]
    rng = np.random.default_rng(7 * p)
    Y = rng.standard_normal((n, p)); U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    prob = R._GenericProblem(At, b, c, n, p)
    prob.y, prob.sigma = y, 0.6
    h = lib.Handle.affine(lib.KIND_GENERIC, At, b, c, n, pcap=p)
    h.set_multipliers(y, 0.6)
    h.set_point(Y)
    f_ref = prob.cost(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-11
    assert _relerr(h.hessvec(U), prob.hess(Y, U)) < 1e-11
    h.close()


@pytest.mark.parametrize("d", [10, 16])
def test_generic_solver_quartic_on_sphere_matches_oracle(lib, d):
    """Full solves (defaults of ManiSDP.m:9-25): GPU and oracle certify the same optimum (KKT residues < 1e-8; the
    reference stores no optimum for these instances)."""
    from manisdp_matlab_amd import solvers
    from oracle import manisdp_ref as R
    At, b, c, K = _qsphere(d)
    Yr, obj_ref, dr = R.ManiSDP(At, b, c, K, {}, verbose=False)
    assert dr["status"] == 0 and max(dr["gap"], dr["pinf"], dr["dinf"]) < 1e-8
    for mode in ("host", "device"):
        Y, obj, data = solvers.ManiSDP(At, b, c, K, {"eig": mode}, verbose=False)
        assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
        assert abs(obj - obj_ref) < 1e-6 * abs(obj_ref)


MC_OPTS = {"tol": 1e-8, "theta": 1e-2, "TR_maxinner": 6, "TR_maxiter": 8, "delta": 10, "alpha": 0.1}   # example_matrixcompletion.m:51-57


def test_matrix_completion_matches_oracle(lib):
    """Nuclear-norm matrix completion through the generic ManiSDP with the options of example/example_matrixcompletion.m
    (the instance generator follows :8-41 with NumPy's random stream): same optimum as the oracle, KKT 1e-8, and the
    four unsampled entries of the rank-2 matrix are recovered (896 of the 900 positions are drawn; with a third of them the
    reference algorithm -- oracle and GPU alike -- ends in its "Slow progress" exit at eta ~ 1e-6)."""
    from manisdp_matlab_amd import problems, solvers
    from oracle import manisdp_ref as R
    At, b, c, K, M, _ = problems.matrix_completion(30, 30, 2, m=6000, seed=3)
    rng = np.random.default_rng(1)
    Y0 = rng.standard_normal((K["s"], 1))
    Yo, objo, do = R.ManiSDP(At, b, c, K, dict(MC_OPTS, Y0=Y0))
    Y, obj, d = solvers.ManiSDP(At, b, c, K, dict(MC_OPTS, Y0=Y0), verbose=False)
    assert do["status"] == 0 and d["status"] == 0
    assert max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    assert abs(obj - objo) <= 1e-7 * abs(objo)
    X12 = Y[:30] @ Y[30:].T
    assert np.linalg.norm(X12 - M) <= 1e-6 * np.linalg.norm(M)


def test_matrix_completion_n1000(lib):
    """p = q = 500, rank 5 (n = 1000, m ~ 200 000 sampled entries): KKT 1e-8, trace = nuclear norm of the completed
    matrix, recovery to 1e-6."""
    from manisdp_matlab_amd import problems, solvers
    At, b, c, K, M, _ = problems.matrix_completion(500, 500, 5, seed=3)
    Y, obj, d = solvers.ManiSDP(At, b, c, K, dict(MC_OPTS), verbose=False, rng=np.random.default_rng(0))
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    X12 = Y[:500] @ Y[500:].T
    assert np.linalg.norm(X12 - M) <= 1e-6 * np.linalg.norm(M)
    assert abs(obj - 2.0 * np.linalg.svd(M, compute_uv=False).sum()) <= 1e-6 * obj      # tr X = 2 |M|_*


SNL_OPTS = {"tol": 1e-4, "sigma0": 1, "sigma_min": 1e1, "theta": 1e-3, "TR_maxiter": 8, "line_search": 0, "alpha": 0.01}   # Sensor_Network_Localization.m:40-46


@pytest.mark.parametrize("n", [4, 10])
def test_sensor_network_localization(lib, n):
    """example/Sensor_Network_Localization.m:2-49 through the generic ManiSDP (the caller SURVEY.md 8f-2 says the generic kind
    unlocks; VERDICT round 4, missing 4): the moment relaxation of the localization quartic with the example's options.  f >= 0
    with f = 0 at the true positions, so the SDP optimum is 0: |optimum| and eta below the example's tolerance; for n = 4 also the
    oracle's optimum from the same start (both 0 to the tolerance)."""
    from manisdp_matlab_amd import problems, solvers
    from oracle import manisdp_ref as R
    f, loc = problems.snl_polynomial(n, seed=1)
    At, b, c, K = problems.snl_mom(f, 2 * n)
    c = np.asarray(c.todense()).ravel()
    maxc = np.abs(c).max()
    Y, fval, d = solvers.ManiSDP(At, b, c / maxc, K, dict(SNL_OPTS), verbose=False, rng=np.random.default_rng(0))
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-4
    assert abs(fval * maxc) < 2e-3
    if n == 4:
        Yo, fo, do = R.ManiSDP(At, b, c / maxc, K, dict(SNL_OPTS), rng=np.random.default_rng(0))
        assert do["status"] == 0 and abs(fo * maxc) < 2e-3
