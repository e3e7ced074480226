"""GPU parity tests for the two affine entry points (ManiSDP_unitdiag / ManiSDP_unittrace):
operators through the C-ABI against the oracle closures on seeded inputs (1e-11 relative:
fp64, SDDMM + MFMA summation orders differ from SciPy/BLAS), then solver-level known answers
from the SDPLIB table the reference ships (tolerance = accuracy the solve certifies)."""
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


def _bqp(d):
    from manisdp_matlab_amd import problems
    Q = np.loadtxt(golden_path(f"bqp_Q_{d}_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path(f"bqp_e_{d}_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(d, Q, e)
    c = np.asarray(c.todense()).ravel()
    return At, b, c / np.abs(c).max(), K


@pytest.mark.parametrize("case,p", [("gpp100", 2), ("gpp100", 7), ("bqp10", 5), ("bqp20", 16), ("gpp124-1", 33), ("gpp100", 140), ("bqp10", 300)])
def test_unitdiag_operators(lib, case, p):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    if case.startswith("bqp"):
        At, b, c, K = _bqp(int(case[3:]))
    else:
        At, b, c, K = problems.from_sdpa(golden_path(case + ".dat-s.gz"))
        c = np.asarray(c.todense()).ravel()
    n = K["s"]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    sigma = 0.37
    prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p)
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n)
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    # co() of the line search at a retracted trial point
    Z = prob.M.retr(Y, 0.5 * U)
    assert abs(h.linesearch_cost(U, 0.5) - prob.cost(Z)) <= 1e-11 * max(1.0, abs(prob.cost(Z)))
    h.close()


@pytest.mark.parametrize("route", ["gram", "sddmm"])
@pytest.mark.parametrize("case,p", [("bqp10", 3), ("bqp20", 32), ("bqp20", 70), ("gpp124-1", 130)])
def test_unitdiag_operator_routes(lib, monkeypatch, route, case, p):
    """A(Ya Yb') through each route (SDDMM per nonzero; Gram matrix by fp64 MFMA + one gather per symmetric pair)
    against the oracle: the library picks the route by bytes moved, here each one is forced."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    monkeypatch.setenv("MSDP_AFFINE_ROUTE", route)
    if case.startswith("bqp"):
        At, b, c, K = _bqp(int(case[3:]))
    else:
        At, b, c, K = problems.from_sdpa(golden_path(case + ".dat-s.gz"))
        c = np.asarray(c.todense()).ravel()
    n = K["s"]
    rng = np.random.default_rng(100 + p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p)
    prob.y, prob.sigma = y, 0.8
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n)
    h.set_multipliers(y, 0.8)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    h.close()


@pytest.mark.parametrize("case,p", [("theta1", 1), ("theta1", 6), ("theta2", 20), ("theta1", 131)])
def test_unittrace_operators(lib, case, p):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    At, b, c, K = problems.from_sdpa(golden_path(case + ".dat-s.gz"))
    c = np.asarray(c.todense()).ravel()
    n = K["s"]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    sigma = 12.5
    prob = R._UnitTraceProblem(At, np.asarray(b, float), c, n, p)
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_UNITTRACE, At, b, c, n)
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    assert _relerr(h.proj(U), prob.M.proj(Y, U)) < 1e-13
    assert _relerr(h.retr(U), prob.M.retr(Y, U)) < 1e-13
    assert _relerr(h.get_point(), Y) == 0.0
    h.close()


@pytest.mark.parametrize("route", ["gram", "sddmm"])
def test_unitdiag_operators_on_slightly_asymmetric_data(lib, monkeypatch, route):
    """SeDuMi data is symmetric and the affine kernels exploit it (upper-triangle adjoint and Gram route).  Data that is
    not exactly symmetric -- here one coefficient of one A_k and one entry of C moved by 1e-13 -- must be detected at
    set-up and take the general kernels (k_adjoint_dense, full Gram matrix); the results stay within the test's
    tolerance of the oracle because the asymmetry itself is far below it."""
    from oracle import manisdp_ref as R
    monkeypatch.setenv("MSDP_AFFINE_ROUTE", route)
    At, b, c, K = _bqp(20)
    n, p = K["s"], 24
    At = At.tocsc(copy=True)
    k = At.shape[1] // 2
    lo, hi = At.indptr[k], At.indptr[k + 1]
    offdiag = [t for t in range(lo, hi) if At.indices[t] % n != At.indices[t] // n]
    At.data[offdiag[0]] *= 1.0 + 1e-13                                 # A_k(i,j) != A_k(j,i)
    c = c.copy(); c[1] += 1e-13                                        # C(2,1) != C(1,2)
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p)
    prob.y, prob.sigma = y, 0.8
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n)
    h.set_multipliers(y, 0.8)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    obj, Ax = h.al_primal(b.size)
    assert _relerr(Ax, prob.A @ (Y @ Y.T).ravel(order="F")) < 1e-11
    h.close()


def test_unitdiag_rtr_single_tcg(lib):
    """maxiter = 1: one tCG; Hess-vec count, stop code and cost agree with the oracle."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    At, b, c, K = _bqp(10)
    n, p = K["s"], 4
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    y = np.zeros(b.size); sigma = 1e-3
    h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n)
    h.set_multipliers(y, sigma)
    for maxinner in (1, 5, 20):
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
        prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p)
        prob.y, prob.sigma = y, sigma
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        assert st.hessvecs == info.hessvecs
        assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-10 * max(1.0, abs(f_ref))
    h.close()


def test_unittrace_rtr_single_tcg(lib):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    c = np.asarray(c.todense()).ravel()
    n, p = K["s"], 3
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
    y = np.zeros(b.size); sigma = 10.0
    h = lib.Handle.affine(lib.KIND_UNITTRACE, At, b, c, n)
    h.set_multipliers(y, sigma)
    for maxinner in (1, 5, 40):
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
        prob = R._UnitTraceProblem(At, np.asarray(b, float), c, n, p)
        prob.y, prob.sigma = y, sigma
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        assert st.hessvecs == info.hessvecs
        assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-10 * max(1.0, abs(f_ref))
    h.close()


def test_solver_bqp_matches_oracle(lib):
    from manisdp_matlab_amd import solvers
    from oracle import manisdp_ref as R
    At, b, c, K = _bqp(10)
    n = K["s"]
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {"Y0": Y0}, verbose=False)
    Yr, objr, datar = R.ManiSDP_unitdiag(At, b, c, K, {"Y0": Y0})
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8 and data["status"] == 0
    assert abs(obj - objr) < 1e-6 * max(1.0, abs(objr))


def test_solver_gpp100_known_answer(lib):
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    At, b, c, K = problems.from_sdpa(golden_path("gpp100.dat-s.gz"))
    opts = dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6)
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, opts, verbose=False)
    eta = max(data["gap"], data["pinf"], data["dinf"])
    assert eta < 1e-6
    assert abs(-obj - known["gpp100"]) < 2e-5 * abs(known["gpp100"])      # README prints 6 digits


def test_solver_theta1_known_answer(lib):
    """theta1 with the options of example/example_theta.m:50-53.  The reference algorithm itself is
    start-point sensitive on this instance: over the seeds 0..5 the oracle certifies for {0, 2, 3}, the GPU path with
    host eig for {1, 2, 4}, with the device escape for {0, 4} (profiles/r2_theta_seeds.log) -- accept/reject and sigma
    decisions at rounding level make the trajectory chaotic, and which starts succeed is not a property the two
    implementations share.  So the test asks that the seeds that converge reproduce the SDPLIB optimum and that at
    least one of three does."""
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    n = K["s"]
    converged = 0
    for seed in (2, 3, 0):
        rng = np.random.default_rng(seed)
        Y0 = rng.standard_normal((n, 1)); Y0 /= np.linalg.norm(Y0)
        opts = dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, Y0=Y0)
        Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, opts, verbose=False)
        eta = max(data["gap"], data["pinf"], data["dinf"])
        assert abs(np.linalg.norm(Y) - 1.0) < 1e-12
        if data["status"] == 0:
            converged += 1
            assert eta < 1e-6
            assert abs(-obj - known["theta1"]) < 1e-5 * known["theta1"]
        else:
            # primal side is still good when the dual certificate stalls
            assert max(data["gap"], data["pinf"]) < 1e-3
    assert converged >= 1


def test_solver_theta2_reaches_the_known_value(lib):
    """theta2 (n = 100, m = 498; data/sdplib/README:99) with the options of example/example_theta.m:50-53.  No run of
    the reference's scheme certifies 1e-6 on this instance -- the oracle and the GPU path end on "Slow progress" at
    eta = 2e-4...5e-4 from every start tried (profiles/r2_theta_seeds.log) -- but every run lands on the SDPLIB optimum
    to 5 digits, which is what this test pins, for the host-eig and the device-escape bookkeeping alike."""
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    At, b, c, K = problems.from_sdpa(golden_path("theta2.dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
    n = K["s"]
    for seed, mode in ((0, "host"), (1, "device")):
        rng = np.random.default_rng(seed)
        Y0 = rng.standard_normal((n, 1)); Y0 /= np.linalg.norm(Y0)
        Y, obj, d = solvers.ManiSDP_unittrace(At, b, c, K, dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, Y0=Y0, eig=mode), verbose=False)
        assert abs(np.linalg.norm(Y) - 1.0) < 1e-12
        assert d["status"] in (0, 2)
        assert abs(-obj - known["theta2"]) <= 5e-5 * known["theta2"]
        assert max(d["gap"], d["pinf"]) < 1e-3


def test_solver_bqp_with_device_escape(lib):
    """ManiSDP_unitdiag with the device few-eigenvector escape (explicit dense S, Lanczos with a GEMV S*v) instead
    of the host eig(S): same certified optimum as the host-eig run."""
    from manisdp_matlab_amd import solvers
    At, b, c, K = _bqp(20)
    Y1, obj1, d1 = solvers.ManiSDP_unitdiag(At, b, c, K, {"eig": "host"}, verbose=False)
    Y2, obj2, d2 = solvers.ManiSDP_unitdiag(At, b, c, K, {"eig": "device"}, verbose=False)
    for d in (d1, d2):
        assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    assert abs(obj1 - obj2) < 1e-6 * max(1.0, abs(obj1))


def test_escape_matrix_matches_lapack(lib):
    from manisdp_matlab_amd import problems
    At, b, c, K = _bqp(20)
    n = K["s"]
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n)); S = (G + G.T) / 2
    Y = rng.standard_normal((n, 4)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n)
    h.set_multipliers(np.zeros(b.size), 1.0)
    h.set_point(Y)
    lam, V, lmax, its = h.escape_eigs_matrix(S, 6, tol=1e-10, maxit=5000)
    h.close()
    dS = np.linalg.eigh(S)[0]
    assert abs(lmax - dS[-1]) < 1e-6 * abs(dS[-1])
    assert np.allclose(lam, dS[:6], rtol=0, atol=1e-8 * abs(dS[0]))
    for t in range(6):
        assert np.linalg.norm(S @ V[:, t] - lam[t] * V[:, t]) < 1e-5 * abs(dS[0])


@pytest.mark.parametrize("kind_name,case,p", [("unitdiag", "gpp100", 6), ("unitdiag", "bqp20", 24), ("unittrace", "theta1", 9),
                                              ("generic", "mcp124-1", 5), ("unitdiag", "bqp10", 140)])
def test_al_bookkeeping_on_device(lib, kind_name, case, p):
    """msdp_al_primal / msdp_al_dual / msdp_escape_eigs_dual against the host expressions of the
    AL loop (ManiSDP_unitdiag.m:59-69, ManiSDP_unittrace.m:59-69, ManiSDP.m:58-66) on seeded inputs."""
    from manisdp_matlab_amd import problems
    if case.startswith("bqp"):
        At, b, c, K = _bqp(int(case[3:]))
    else:
        At, b, c, K = problems.from_sdpa(golden_path(case + ".dat-s.gz"))
        c = np.asarray(c.todense()).ravel()
    b = np.asarray(b, float)
    n = K["s"]
    kind = {"unitdiag": lib.KIND_UNITDIAG, "unittrace": lib.KIND_UNITTRACE, "generic": lib.KIND_GENERIC}[kind_name]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p))
    if kind_name == "unitdiag":
        Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    elif kind_name == "unittrace":
        Y /= np.linalg.norm(Y)
    y = rng.standard_normal(b.size) * 0.3
    h = lib.Handle.affine(kind, At, b, c, n, pcap=p)
    h.set_multipliers(np.zeros(b.size), 1.0)
    h.set_point(Y)
    X = Y @ Y.T
    x = X.ravel(order="F")
    obj, Ax = h.al_primal(b.size)
    assert abs(obj - float(c @ x)) <= 1e-11 * max(1.0, abs(float(c @ x)))
    assert _relerr(Ax, At.T @ x) < 1e-11
    eS = (c - At @ y).reshape((n, n), order="F")
    z = h.al_dual(y)
    if kind_name == "unitdiag":
        z_ref = np.sum(X * eS, axis=0)
        S = eS - np.diag(z_ref)
        assert _relerr(z, z_ref) < 1e-11
    elif kind_name == "unittrace":
        z_ref = float(np.sum(eS * X))
        S = eS - z_ref * np.eye(n)
        assert abs(z - z_ref) <= 1e-11 * max(1.0, abs(z_ref))
    else:
        S = eS
        assert z is None
    S = 0.5 * (S + S.T)
    dS = np.linalg.eigvalsh(S)
    lam, V, lmax, _ = h.escape_eigs_dual(3, tol=1e-10, maxit=20000)
    scale = max(abs(dS[0]), abs(dS[-1]))
    assert abs(lam[0] - dS[0]) < 1e-7 * scale
    assert abs(lmax - dS[-1]) < 1e-5 * scale
    # the point is still resident and usable after the bookkeeping calls
    assert _relerr(h.get_point(), Y) == 0.0
    f0 = h.cost()
    assert np.isfinite(f0)
    h.close()


def test_solvers_with_device_al_bookkeeping_all_kinds(lib):
    """eig='device' routes the whole AL step of the three affine-type entry points through msdp_al_primal / _dual /
    msdp_escape_eigs_dual; each must land on the optimum the host-bookkeeping run (eig='host') certifies."""
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    # generic entry point on mcp100 (known answer)
    At, b, c, K = problems.from_sdpa(golden_path("mcp100.dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
    Y, obj, d = solvers.ManiSDP(At, b, c, K, {"eig": "device"}, verbose=False, rng=np.random.default_rng(0))
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    assert abs(-obj - known["mcp100"]) < 1e-6 * known["mcp100"]
    # unit diagonal on gpp100 (known answer, options of the oracle test)
    At, b, c, K = problems.from_sdpa(golden_path("gpp100.dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
    opts = dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6, eig="device")
    Y, obj, d = solvers.ManiSDP_unitdiag(At, b, c, K, opts, verbose=False)
    assert max(d["gap"], d["pinf"], d["dinf"]) < 1e-6
    assert abs(-obj - known["gpp100"]) < 2e-5 * abs(known["gpp100"])
    # unit trace on theta1: same start point with host and device bookkeeping
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
    rng = np.random.default_rng(3)
    Y0 = rng.standard_normal((K["s"], 1)); Y0 /= np.linalg.norm(Y0)
    res = []
    for mode in ("host", "device"):
        Y, obj, d = solvers.ManiSDP_unittrace(At, b, c, K, dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, Y0=Y0, eig=mode), verbose=False)
        res.append((obj, max(d["gap"], d["pinf"]), d["status"]))
        assert abs(np.linalg.norm(Y) - 1.0) < 1e-12
    # The primal side agrees.  With these options the dual certificate of this instance may stall for either mode (see
    # the theta1 test), and where a stalled run stops moves with the last bits of the Hess-vec sums: a run that
    # certifies (status 0) must hit the known optimum, a stalled one (status 2) stays within 1 % of it.
    assert max(res[0][1], res[1][1]) < 1e-3
    for obj, _, status in res:
        assert abs(-obj - known["theta1"]) < (1e-5 if status == 0 else 1e-2) * known["theta1"]


def test_theta1_with_the_reference_examples_options(lib):
    """example_theta.m:48-55 does not run ManiSDP_unittrace with its defaults but with tol = 1e-6, sigma0 = 1e5,
    sigma_max = 1e8 and the line search.  With those options the solve converges (status 0, eta < 1e-6) for about half of
    the start points -- the oracle: 2 of 3 -- and every converged run hits SDPLIB's value 23 (data/sdplib/README:98) to
    1e-6 relative, the tolerance north_star states."""
    from manisdp_matlab_amd import problems, solvers
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    opts = {"tol": 1e-6, "sigma0": 1e5, "sigma_max": 1e8, "line_search": 1}
    converged = 0
    for seed in range(6):
        _, obj, d = solvers.ManiSDP_unittrace(At, b, c, K, dict(opts), rng=np.random.default_rng(seed), verbose=False)
        if d["status"] == 0:
            converged += 1
            assert max(d["gap"], d["pinf"], d["dinf"]) < 1e-6
            assert abs(obj + 23.0) <= 1e-6 * 23.0, (seed, obj)
    assert converged >= 2, converged


@pytest.mark.parametrize("kind_name,case,p,shape", [("unitdiag", "bqp20", 16, 1), ("unitdiag", "bqp20", 32, 2), ("unitdiag", "gpp124-1", 9, 3),
                                                   ("unittrace", "theta2", 20, 1), ("unittrace", "theta1", 6, 2), ("generic", "mcp124-1", 5, 3)])
def test_affine_operators_through_the_symmetric_contraction(lib, kind_name, case, p, shape):
    """The dense products of the affine closures (C*Y, eS*Y, 2*eS*U + 4*sigma*AyU*Y: ManiSDP_unitdiag.m:160-170,
    ManiSDP_unittrace.m:160-176, ManiSDP.m:153-162) through msdp_densesym.hip -- forced at these small sizes with option
    dense_sym = 2, one- and two-matrix launches, every workgroup shape -- against the oracle."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    if case.startswith("bqp"):
        At, b, c, K = _bqp(int(case[3:]))
    else:
        At, b, c, K = problems.from_sdpa(golden_path(case + ".dat-s.gz"))
        c = np.asarray(c.todense()).ravel()
    n = K["s"]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p))
    if kind_name == "unitdiag":
        Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p); kind = lib.KIND_UNITDIAG
    elif kind_name == "unittrace":
        Y /= np.linalg.norm(Y)
        prob = R._UnitTraceProblem(At, np.asarray(b, float), c, n, p); kind = lib.KIND_UNITTRACE
    else:
        prob = R._GenericProblem(At, np.asarray(b, float), c, n, p); kind = lib.KIND_GENERIC
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(np.asarray(b).size) * 0.1
    sigma = 0.9
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(kind, At, b, c, n)
    h.set_option("dense_sym", 2); h.set_option("dense_sym_rt", shape)
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    H = h.hessvec(U)
    assert _relerr(H, H_ref) < 1e-11
    assert np.array_equal(H, h.hessvec(U))
    h.close()


@pytest.mark.parametrize("case,p", [("bqp10", 3), ("bqp20", 32), ("bqp30", 20), ("bqp20", 70)])
def test_unitdiag_hessvec_b_route_matches_the_two_pass_route(lib, monkeypatch, case, p):
    """Hess-vec of ManiSDP_unitdiag.m:166-171 on the Gram route with A'(A(.)) applied as ONE sparse matrix to Y'U + U'Y
    (k_adjoint_gram, option affine_broute, default) against the two-pass form (k_gram_apply -> w -> k_adjoint_tiled) and the oracle."""
    from oracle import manisdp_ref as R
    monkeypatch.setenv("MSDP_AFFINE_ROUTE", "gram")
    At, b, c, K = _bqp(int(case[3:]))
    n = K["s"]
    rng = np.random.default_rng(7 + p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(b.size) * 0.1
    prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p)
    prob.y, prob.sigma = y, 1.7
    prob.cost(Y); prob.grad(Y)                                # the closures share Axb and eS (ManiSDP_unitdiag.m:152-164)
    H_ref = prob.hess(Y, U)
    out = []
    for broute in (1, 0):
        h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n)
        h.set_option("affine_broute", broute)
        h.set_multipliers(y, 1.7)
        h.set_point(Y)
        out.append(h.hessvec(U))
        assert _relerr(out[-1], H_ref) < 1e-11
        h.close()
    assert _relerr(out[0], out[1]) < 1e-12


@pytest.mark.parametrize("kind_name,case,p", [("unittrace", "theta1", 6), ("unittrace", "theta3", 17), ("unittrace", "theta3", 140),
                                              ("unitdiag", "gpp100", 7), ("unitdiag", "gpp124-1", 33), ("generic", "mcp124-1", 5),
                                              ("generic", "theta3", 12)])
def test_fused_sddmm_and_sphere_epilogue_match_the_separate_launches(lib, monkeypatch, kind_name, case, p):
    """Option affine_fuse (default 1): A(Ya Yb') and its finish in one launch (k_sddmm1; long constraints -- the trace row of
    theta3, the all-ones constraint of gpp -- summed by the workgroup that arrives last), and for the sphere / Euclidean factor
    the sparse A'(w)*Y product, the slab sum and the projection in one launch with tr(H Y') assembled from <U,G>, <U,Y> and
    <w, A(YY')> (k_sph_hess_fused; ManiSDP_unittrace.m:171-177, ManiSDP.m:158-162).  Against the separate launches and the oracle."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    monkeypatch.setenv("MSDP_AFFINE_ROUTE", "sddmm")
    At, b, c, K = problems.from_sdpa(golden_path(case + ".dat-s.gz"))
    c = np.asarray(c.todense()).ravel()
    b = np.asarray(b, float).ravel()
    n = K["s"]
    rng = np.random.default_rng(3 * p)
    Y = rng.standard_normal((n, p))
    if kind_name == "unitdiag":
        Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        prob = R._UnitDiagProblem(At, b, c, n, p); kind = lib.KIND_UNITDIAG
    elif kind_name == "unittrace":
        Y /= np.linalg.norm(Y)
        prob = R._UnitTraceProblem(At, b, c, n, p); kind = lib.KIND_UNITTRACE
    else:
        prob = R._GenericProblem(At, b, c, n, p); kind = lib.KIND_GENERIC
    U = rng.standard_normal((n, p))
    if kind_name != "generic":
        U = prob.M.proj(Y, U)                                  # tCG only ever multiplies tangent vectors
    y = rng.standard_normal(b.size) * 0.1
    sigma = 2.3
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    out = []
    for fuse, side in ((1, 1), (0, 0), (1, 0)):                # side: the SDDMM as a side job of the contraction launch (sphere, few touched entries)
        h = lib.Handle.affine(kind, At, b, c, n)
        h.set_option("affine_fuse", fuse); h.set_option("affine_side", side)
        h.set_multipliers(y, sigma)
        h.set_point(Y)
        f, G, H = h.cost(), h.rgrad(), h.hessvec(U)
        assert abs(f - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
        assert _relerr(G, G_ref) < 1e-11
        assert _relerr(H, H_ref) < 1e-11
        assert np.array_equal(H, h.hessvec(U))               # the arrival counter is back at zero, same bits
        out.append((f, G, H))
        h.close()
    for q in (1, 2):
        assert abs(out[0][0] - out[q][0]) <= 1e-13 * max(1.0, abs(f_ref))
        assert _relerr(out[0][1], out[q][1]) < 1e-13
        assert _relerr(out[0][2], out[q][2]) < 1e-12


@pytest.mark.parametrize("kind_name,p", [("unittrace", 6), ("unittrace", 40), ("generic", 9)])
def test_sphere_hessvec_with_more_long_constraints_than_waves(lib, kind_name, p):
    """k_sph_hess_fused kept the values of the long constraints (more than 128 nonzeros) in an LDS array of
    MSDP_WAVES entries; only the side-job route was guarded by nlong <= MSDP_WAVES.  24 long constraints + the trace row, few
    touched entries: every route (side job refused -> k_sddmm1 + fused epilogue, separate launches) against the oracle
    (ManiSDP_unittrace.m:171-177, ManiSDP.m:158-162)."""
    import scipy.sparse as sp
    from oracle import manisdp_ref as R
    n, nlong, nshort = 320, 24, 150
    rng = np.random.default_rng(11)
    cols = []
    for k in range(nlong):
        ii = rng.integers(0, n, 400); jj = rng.integers(0, n, 400)
        keep = ii < jj
        A = sp.coo_matrix((rng.choice([1.0, -0.5, 0.25], keep.sum()), (ii[keep], jj[keep])), shape=(n, n)).tocsr()
        A = A + A.T
        assert A.nnz > 2 * 128
        cols.append(A.reshape((n * n, 1)).tocsc())
    for k in range(nshort):
        i, j = sorted(rng.choice(n, 2, replace=False))
        A = sp.coo_matrix(([1.0, 1.0], ([i, j], [j, i])), shape=(n, n))
        cols.append(A.reshape((n * n, 1)).tocsc())
    cols.append(sp.identity(n, format="csr").reshape((n * n, 1)).tocsc())          # the trace row, last (example_theta.m:36-39)
    At = sp.hstack(cols).tocsc()
    m = At.shape[1]
    b = np.zeros(m); b[-1] = 1.0
    G0 = rng.standard_normal((n, n)); c = ((G0 + G0.T) / (2 * np.sqrt(n))).ravel()
    Y = rng.standard_normal((n, p))
    if kind_name == "unittrace":
        Y /= np.linalg.norm(Y)
        prob = R._UnitTraceProblem(At, b, c, n, p); kind = lib.KIND_UNITTRACE
    else:
        prob = R._GenericProblem(At, b, c, n, p); kind = lib.KIND_GENERIC
    U = rng.standard_normal((n, p))
    if kind_name != "generic":
        U = prob.M.proj(Y, U)
    y = rng.standard_normal(m) * 0.1
    sigma = 3.7
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    for fuse, side in ((1, 1), (1, 0), (0, 0)):
        h = lib.Handle.affine(kind, At, b, c, n)
        h.set_option("affine_fuse", fuse); h.set_option("affine_side", side)
        h.set_multipliers(y, sigma)
        h.set_point(Y)
        assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
        assert _relerr(h.rgrad(), G_ref) < 1e-11
        H = h.hessvec(U)
        assert _relerr(H, H_ref) < 1e-11, (fuse, side)
        assert np.array_equal(H, h.hessvec(U))
        h.close()
