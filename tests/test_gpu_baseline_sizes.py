"""GPU parity at the sizes BASELINE.json's configs name ("configs_untested").

 * config 3: BQP d = 60 (data/bqp_Q_60_1.txt of the reference -> n = 1831, m = 1 155 281, nnz(At) = 4.8 M) through
   ManiSDP_unitdiag: operators against the oracle on both A(Ya Yb') routes, and the full solve from the default start;
 * config 4: a unit-trace, dense-C problem at n = 5000 (example_theta.m scaled up) -- operators against the oracle --
   and the named workload quartic-on-the-sphere at d = 60 (n = 1891, example_qsphere.m:18-27) solved to KKT 1e-8;
 * config 5: the per-GPU shard of the synthetic dense n = 100 000, p = 64 problem at FULL size (12 500 x 100 000 rows):
   sampled rows of eG and of the Hess-vec recomputed on the host from the generator, plus linearity.

The oracle cannot run whole solves of these sizes inside a test (BQP d = 60: 311 s, quartic d = 60: minutes), so the
optima it certifies are committed in tests/golden/oracle_optima.json (made by tests/golden/make_oracle_optima.py in
the build container) and the solves here must reach the same value AND certify KKT < 1e-8 themselves."""
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


@pytest.fixture(scope="module")
def optima():
    return json.load(open(golden_path("oracle_optima.json")))


@pytest.fixture(scope="module")
def bqp60():
    from manisdp_matlab_amd import problems
    Q = np.loadtxt(golden_path("bqp_Q_60_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path("bqp_e_60_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(60, Q, e)
    c = np.asarray(c.todense()).ravel()
    return At, np.asarray(b, float), c / np.abs(c).max(), K          # example_bqp.m:39-41 scaling


# --------------------------------------------------------------------------------------------- config 3
@pytest.mark.parametrize("route,p", [("gram", 16), ("sddmm", 16), ("gram", 32), ("sddmm", 32), ("gram", 300)])
def test_bqp60_operators(lib, bqp60, monkeypatch, route, p):
    from oracle import manisdp_ref as R
    monkeypatch.setenv("MSDP_AFFINE_ROUTE", route)
    At, b, c, K = bqp60
    n = K["s"]
    assert (n, b.size) == (1831, 1155281)                             # data/bqp_result.txt:8 of the reference
    rng = np.random.default_rng(60 + p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(b.size)
    sigma = 0.04
    prob = R._UnitDiagProblem(At, b, c, n, p)
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_UNITDIAG, At, b, c, n, pcap=p)
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    # AL bookkeeping kernels at the same size (ManiSDP_unitdiag.m:59-67)
    obj, Ax = h.al_primal(b.size)
    x = (Y @ Y.T).ravel(order="F")
    assert abs(obj - c @ x) <= 1e-11 * max(1.0, abs(c @ x))
    assert _relerr(Ax, prob.A @ x) < 1e-11
    z = h.al_dual(y)
    eS = (c - prob.At @ y).reshape((n, n), order="F")
    assert _relerr(z, np.sum((Y @ Y.T) * eS, axis=0)) < 1e-11
    assert _relerr(h.get_dual_slack(), eS - np.diag(z)) < 1e-11
    h.close()


def test_bqp60_solve_default_start(lib, bqp60, optima):
    """example_bqp.m:36-43 with default options from the default start (rng seed 0, p0 = 2): the solve must end with
    "Optimality is reached" -- status 0, eta < 1e-8 -- at the optimum the oracle certifies for this instance."""
    from manisdp_matlab_amd import solvers
    At, b, c, K = bqp60
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False)
    eta = max(data["gap"], data["pinf"], data["dinf"])
    assert data["status"] == 0, (obj, eta, data["iters"])
    assert eta < 1e-8
    assert abs(obj - optima["bqp60_1"]["obj"]) <= 1e-6 * abs(optima["bqp60_1"]["obj"])
    # independent feasibility check on the host: A(YY') = b, unit diagonal
    X = Y @ Y.T
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) / (1 + np.linalg.norm(b)) < 1e-8
    assert np.abs(np.diag(X) - 1).max() < 1e-12


# --------------------------------------------------------------------------------------------- config 4
@pytest.mark.parametrize("p", [8, 32])
def test_unittrace_n5000_operators(lib, p):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    n = 5000
    At, b, c, K = problems.theta_problem(n, ndraws=10 * n, seed=1)    # example_theta.m:2-39 at n = 5000
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
    U = rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(b.size)
    sigma = 1e3
    prob = R._UnitTraceProblem(At, b, c, n, p)
    prob.y, prob.sigma = y, sigma
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    h = lib.Handle.affine(lib.KIND_UNITTRACE, At, b, c, n, pcap=p)
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-11
    assert _relerr(h.hessvec(U), H_ref) < 1e-11
    assert _relerr(h.proj(U), prob.M.proj(Y, U)) < 1e-13
    assert _relerr(h.retr(U), prob.M.retr(Y, U)) < 1e-13
    h.close()


def test_qsphere60_solve(lib, optima):
    """Quartic on the sphere at d = 60 (n = 1891, m = 1 155 402) through the generic ManiSDP as example_qsphere.m:18-27
    does: converges (status 0, eta < 1e-8) to the optimum the oracle certifies."""
    from manisdp_matlab_amd import problems as P, solvers
    d = 60
    coe = np.random.default_rng(5).standard_normal(P.get_basis(d, 4).shape[1])
    At, b, c, K = P.qsmom(d, coe)
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    assert K["s"] == 1891
    Y, obj, data = solvers.ManiSDP(At, b, c, K, {}, verbose=False)
    eta = max(data["gap"], data["pinf"], data["dinf"])
    assert data["status"] == 0 and eta < 1e-8, (obj, eta, data["iters"])
    assert abs(obj - optima["qsphere60_seed5"]["obj"]) <= 1e-6 * abs(optima["qsphere60_seed5"]["obj"])


# --------------------------------------------------------------------------------------------- config 5
def _syn_rows(n, rows, seed):
    """Rows of the synthetic dense C (msdp_synthetic_dense_entry: splitmix64 of min(i,j)*n + max(i,j)) in NumPy."""
    rows = np.asarray(rows, dtype=np.uint64)[:, None]
    cols = np.arange(n, dtype=np.uint64)[None, :]
    a = np.minimum(rows, cols); bb = np.maximum(rows, cols)
    g = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        x = a * np.uint64(n) + bb + np.uint64(seed) * g
        x = x + g
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (2.0 * u - 1.0) / np.sqrt(float(n))


@pytest.mark.parametrize("rank", [0, 7])
def test_config5_shard_full_size(lib, rank):
    n, p, N, seed = 100000, 64, 8, 0
    L = lib.load()
    # the NumPy restatement of the generator is the library's host function
    probe = [(0, 0), (3, 99999), (99999, 3), (12499, 12500), (54321, 777)]
    R = _syn_rows(n, [i for i, _ in probe], seed)
    for q, (i, j) in enumerate(probe):
        assert R[q, j] == L.msdp_synthetic_dense_entry(n, i, j, seed)
    h = lib.Handle.dense_synthetic(n, seed, nranks=N, rank=rank, pcap=p)
    r0, r1 = h.local_rows()
    assert r1 - r0 == 12500
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p)); U -= Y * np.sum(Y * U, axis=1, keepdims=True)
    V = rng.standard_normal((n, p)); V -= Y * np.sum(Y * V, axis=1, keepdims=True)
    h.set_point(Y)
    h.debug_set_full_rows(Y)
    z = h.get_z()                                                     # eG = sum((Y*C).*Y) on the local rows
    G = h.rgrad()
    h.debug_set_full_rows(U)
    HU = h.hessvec(U)
    h.debug_set_full_rows(V)
    HV = h.hessvec(V)
    W = 0.7 * U - 1.3 * V
    h.debug_set_full_rows(W)
    HW = h.hessvec(W)
    h.close()
    # 64 sampled local rows (first, last, and random ones) recomputed on the host from the generator
    rows = np.unique(np.concatenate([[r0, r0 + 1, r1 - 1], rng.integers(r0, r1, size=64)]))[:64 + 3]
    Crows = _syn_rows(n, rows, seed)                                  # 67 x 100000
    CY = Crows @ Y
    eG = np.sum(CY * Y[rows], axis=1)
    assert _relerr(z[rows], eG) < 1e-12
    assert _relerr(G[rows], CY - Y[rows] * eG[:, None]) < 1e-12       # ManiSDP_onlyunitdiag.m:124
    CU = Crows @ U
    H_ref = CU - Y[rows] * np.sum(Y[rows] * CU, axis=1, keepdims=True) - U[rows] * eG[:, None]    # :128-129
    assert _relerr(HU[rows], H_ref) < 1e-12
    # size-independent properties on ALL local rows: linearity of the Hess-vec, tangency of the result
    assert _relerr(HW[r0:r1], 0.7 * HU[r0:r1] - 1.3 * HV[r0:r1]) < 1e-12
    assert np.abs(np.sum(HU[r0:r1] * Y[r0:r1], axis=1)).max() < 1e-11
