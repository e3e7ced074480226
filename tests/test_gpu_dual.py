"""GPU parity tests for the dual approach with a unit-diagonal dual slack (MSDP_KIND_DUAL_UNITDIAG; reference
src/dual/ManiDSDP_unitdiag.m) through the C ABI, against the oracle's restatement of the same file.
Tolerances: operators 1e-11 relative (fp64, different summation orders); full solves reach the optimum of the
primal moment relaxation of the same BQP (strong duality) to 1e-6 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


def _bqp(d, seed):
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
    e = rng.standard_normal(d)
    return Q, e


def _dual_data(d, seed):
    from manisdp_matlab_amd import problems
    Q, e = _bqp(d, seed)
    A, b, c, K, dAAt, maxb = problems.bqpsos_dual_problem(Q, e, d)
    return Q, e, A, b, c, K, dAAt, maxb


def _oracle_problem(A, b, c, K, dAAt, p):
    import scipy.sparse as sp
    from oracle import manisdp_ref as R
    nf, n = K["f"], K["s"]
    Ac = sp.csc_matrix(A)
    return R._DualUnitDiagProblem(Ac[:, nf:], Ac[:, :nf], b, c[nf:], c[:nf], dAAt, n, p)


@pytest.mark.parametrize("d,p", [(6, 5), (9, 12), (12, 33)])
def test_dual_operators_match_oracle(lib, d, p):
    import scipy.sparse as sp
    _, _, A, b, c, K, dAAt, _ = _dual_data(d, seed=d)
    nf, n = K["f"], K["s"]
    prob = _oracle_problem(A, b, c, K, dAAt, p)
    Ac = sp.csc_matrix(A)
    h = lib.Handle.dual_unitdiag(sp.csr_matrix(Ac[:, nf:]), b, c[nf:], dAAt, Ac[:, :nf], c[:nf], pcap=max(32, p))
    rng = np.random.default_rng(p)
    def point():
        Y = rng.standard_normal((n, p)); return Y / np.linalg.norm(Y, axis=1, keepdims=True)
    # outer step at a random point with sigma0: moves the device-resident multiplier x away from 0
    sigma0, w0 = 0.37, np.array([0.2])
    Y0 = point()
    h.dual_set_penalty(sigma0, w0)
    h.set_point(Y0)
    by, cex, as2, Af, z = h.dual_outer_step()
    S, sc, y = prob.parts(Y0)
    As = prob.At @ y - sc
    Af_ref = prob.B.T @ y - prob.cf
    assert abs(by - b @ y) <= 1e-11 * max(1.0, abs(b @ y))
    assert abs(as2 - As @ As) <= 1e-11 * max(1.0, As @ As)
    assert np.max(np.abs(Af - Af_ref)) < 1e-11 * (1.0 + np.linalg.norm(y))     # B'y - cf is 0 up to rounding here (y_1 = tr(S)/n = 1)
    prob.x = prob.x - sigma0 * As
    eX = (prob.x + prob.bA).reshape((n, n), order="F")
    z_ref = np.sum(S * eX, axis=0)
    assert _relerr(z, z_ref) < 1e-11
    assert abs(cex - prob.c @ eX.ravel(order="F")) <= 1e-11 * max(1.0, np.abs(eX).sum())
    assert _relerr(h.get_dual_slack(), eX - np.diag(z_ref)) < 1e-11
    assert _relerr(h.dual_get_y(), y) < 1e-11
    # cost / gradient / Hess-vec at another point with other multipliers
    sigma, w = 2.3, np.array([-0.7])
    prob.sigma, prob.w = sigma, w
    Y, U = point(), rng.standard_normal((n, p))
    h.dual_set_penalty(sigma, w)
    h.set_point(Y)
    f_ref = prob.cost(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-11
    Ut = prob.M.proj(Y, U)
    assert _relerr(h.hessvec(Ut), prob.hess(Y, Ut)) < 1e-11
    # co() of the line search at retr(Y + alpha*V)
    V = rng.standard_normal((n, p))
    Yt = Y + 0.3 * V; Yt /= np.linalg.norm(Yt, axis=1, keepdims=True)
    assert abs(h.linesearch_cost(V, 0.3) - prob.cost(Yt)) <= 1e-11 * max(1.0, abs(prob.cost(Yt)))
    h.close()


@pytest.mark.parametrize("maxinner", [1, 3, 20])
def test_dual_single_rtr_matches_oracle(lib, maxinner):
    """One trustregions() call: same Hess-vec count, cost and gradient norm as the oracle's RTR on the same closures."""
    import scipy.sparse as sp
    from oracle.manopt_rtr import trustregions
    d, p = 8, 6
    _, _, A, b, c, K, dAAt, _ = _dual_data(d, seed=3)
    nf, n = K["f"], K["s"]
    prob = _oracle_problem(A, b, c, K, dAAt, p)
    prob.sigma = 1e-3
    Ac = sp.csc_matrix(A)
    h = lib.Handle.dual_unitdiag(sp.csr_matrix(Ac[:, nf:]), b, c[nf:], dAAt, Ac[:, :nf], c[:nf])
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h.dual_set_penalty(1e-3, np.zeros(1))
    h.set_point(Y)
    st = h.rtr(lib.default_opts(maxiter=4, maxinner=maxinner, tolgradnorm=1e-8))
    Yr, fr, info = trustregions(prob, Y.copy(), 4, maxinner, 1e-8)
    assert st.hessvecs == info.hessvecs
    assert abs(st.cost - fr) <= 1e-10 * max(1.0, abs(fr))
    assert abs(st.gradnorm - info.gradnorm) <= 1e-8 * max(1.0, info.gradnorm)
    assert _relerr(h.get_point(), Yr) < 1e-8
    h.close()


@pytest.mark.parametrize("d,line_search", [(8, 1), (10, 0), (14, 1)])
def test_dual_solve_reaches_primal_optimum(lib, d, line_search):
    """example/dual/example_bqp_dual.m at small d: the dual approach on the SOS relaxation and the primal approach on the
    moment relaxation of the same BQP end at the same value (strong duality); the oracle's dual solve agrees too."""
    from manisdp_matlab_amd import problems, solvers
    from oracle import manisdp_ref as R
    Q, e, A, b, c, K, dAAt, maxb = _dual_data(d, seed=100 + d)
    At, bp, cp, Kp = problems.bqpmom(d, Q, e)
    _, fprimal, dp = R.ManiSDP_unitdiag(At, bp, cp, Kp, {"tol": 1e-8}, rng=np.random.default_rng(0))
    assert max(dp["gap"], dp["pinf"], dp["dinf"]) < 1e-8
    n = K["s"]; p0 = int(np.ceil(np.log(b.size)))
    Y0 = np.random.default_rng(5).standard_normal((n, p0)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    opts = {"tol": 1e-8, "dAAt": dAAt, "line_search": line_search, "Y0": Y0}
    X, obj, data = solvers.ManiDSDP_unitdiag(A, b, c, K, dict(opts), verbose=False)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert abs(obj * maxb - fprimal) <= 1e-6 * max(1.0, abs(fprimal))
    Xo, objo, datao = R.ManiDSDP_unitdiag(A, b, c, K, dict(opts))
    assert abs(obj - objo) <= 1e-7 * max(1.0, abs(objo))
    # first outer iterations follow the oracle's (same start point): obj, pinf, p
    for k in range(min(3, len(data["log"]), len(datao["log"]))):
        g, r = data["log"][k], datao["log"][k]
        assert abs(g[0] - r[0]) <= 1e-6 * max(1.0, abs(r[0])), (k, g, r)
        assert g[6] == r[6]
    # X = eX - diag(z) (:81) satisfies every constraint of A(X) + B(w) = b whose matrix has a zero diagonal (all but
    # the trace row, which sees the diagonal multiplier z)
    import scipy.sparse as sp
    nf = K["f"]
    Ac = sp.csr_matrix(A)
    res = Ac[:, nf:] @ X.ravel(order="F") + Ac[:, :nf] @ data["w"] - b
    assert np.linalg.norm(res[1:]) / (1 + np.linalg.norm(b)) < 1e-6


def test_dual_eig_device_mode(lib):
    """The same solve with the device escape (Lanczos on the resident X) instead of the host eig(X)."""
    from manisdp_matlab_amd import solvers
    d = 10
    Q, e, A, b, c, K, dAAt, maxb = _dual_data(d, seed=42)
    o = {"tol": 1e-8, "dAAt": dAAt, "line_search": 1}
    _, obj_h, dh = solvers.ManiDSDP_unitdiag(A, b, c, K, dict(o, eig="host"), verbose=False)
    _, obj_d, dd = solvers.ManiDSDP_unitdiag(A, b, c, K, dict(o, eig="device"), verbose=False)
    assert dh["status"] == 0 and dd["status"] == 0
    assert abs(obj_h - obj_d) <= 1e-7 * max(1.0, abs(obj_h))


def test_dual_example_size_d30(lib):
    """example/dual/example_bqp_dual.m:3-37 at its own size (d = 30: n = 466, m = 31 931, line search on): the device
    solve reaches the tolerance and the oracle's optimum of the same SOS relaxation."""
    from manisdp_matlab_amd import solvers
    from oracle import manisdp_ref as R
    d = 30
    Q, e, A, b, c, K, dAAt, maxb = _dual_data(d, seed=1)
    assert K["s"] == 466 and b.size == 31931
    o = {"tol": 1e-8, "dAAt": dAAt, "line_search": 1}
    _, obj, data = solvers.ManiDSDP_unitdiag(A, b, c, K, dict(o), verbose=False)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    _, objo, datao = R.ManiDSDP_unitdiag(A, b, c, K, dict(o))
    assert datao["status"] == 0
    assert abs(obj - objo) * maxb <= 1e-6 * max(1.0, abs(objo) * maxb)


def test_dual_without_free_variables_and_width_limit(lib):
    """K.f = 0 (no free block: B, c_f, w absent) against the oracle's closures, and the documented width limit of the
    dual kind (p <= 128) reported as an error, not a wrong answer."""
    import scipy.sparse as sp
    from oracle import manisdp_ref as R
    _, _, A, b, c, K, dAAt, _ = _dual_data(7, seed=11)
    n = K["s"]
    Apsd = sp.csr_matrix(sp.csc_matrix(A)[:, 1:])
    rng = np.random.default_rng(0)
    cpsd = rng.standard_normal((n, n)); cpsd = 0.1 * (cpsd + cpsd.T)
    p = 9
    prob = R._DualUnitDiagProblem(Apsd, sp.csr_matrix((b.size, 0)), b, cpsd.ravel(order="F"), np.zeros(0), dAAt, n, p)
    prob.sigma = 0.8
    h = lib.Handle.dual_unitdiag(Apsd, b, cpsd.ravel(order="F"), dAAt, None, None)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = prob.M.proj(Y, rng.standard_normal((n, p)))
    h.dual_set_penalty(0.8)
    h.set_point(Y)
    f_ref = prob.cost(Y)
    assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-11
    assert _relerr(h.hessvec(U), prob.hess(Y, U)) < 1e-11
    by, cex, as2, Af, z = h.dual_outer_step()
    assert Af.size == 0 and abs(by - b @ prob.parts(Y)[2]) < 1e-11 * max(1.0, abs(by))
    # width limit
    Yw = rng.standard_normal((n, 140)); Yw /= np.linalg.norm(Yw, axis=1, keepdims=True)
    h.dual_set_penalty(0.8)
    h.set_point(Yw)
    with pytest.raises(lib.MsdpError, match="exceeds the supported maximum"):
        h.cost()
    h.close()
