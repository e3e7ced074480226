"""Independent checks of what the AL loops certify (ADVICE: unconverged Lanczos).

 * G81 (BASELINE config 2) solved to dinf < 1e-8 by ManiSDP_onlyunitdiag with the device escape: lambda_min(S) and
   lambda_max(S) of S = C - diag(z) (ManiSDP_onlyunitdiag.m:45-51) are recomputed on the host with ARPACK in
   shift-invert mode (sparse LU of S - sigma I) and must give the same dinf -- for the reference example's p0 = 40
   (example_maxcut.m:32) and for the default p0 = 2;
 * msdp_escape_info: a Lanczos run that hits `maxit` is reported as not converged, missing pairs come back as +inf
   with zero vectors, and the AL loop does not declare optimality on an uncertified dinf;
 * a13: co() and line_search() of ManiSDP_onlyunitdiag.m:99-115 against the oracle, operator level and solver level."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


def _neg_count(S, mu):
    """Number of eigenvalues of the sparse symmetric S below mu, by Sylvester's law of inertia: S - mu*I is factorised
    WITHOUT row interchanges (SuperLU, symmetric mode, pivot threshold 0), so P(S - mu I)P' = L U with U = D L' and the
    signs of diag(U) are the signs of the eigenvalues."""
    import scipy.sparse.linalg as spla
    A = (S - mu * sp.identity(S.shape[0], format="csc")).tocsc()
    lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    assert np.array_equal(lu.perm_r, lu.perm_c)
    return int(np.sum(lu.U.diagonal() < 0))


def _host_dinf(C, Y):
    """dinf of ManiSDP_onlyunitdiag.m:45-51 recomputed on the host, independently of any Krylov process: lambda_max by
    ARPACK (an easy, isolated eigenvalue), lambda_min by bisection on the inertia of S - mu*I (a sparse LDL'-type
    factorisation per step; 0.1 s for the 20000 x 20000 grid graph)."""
    import scipy.sparse.linalg as spla
    z = np.asarray(np.sum((C @ Y) * Y, axis=1)).ravel()              # :46-47
    S = (C - sp.diags(z)).tocsc()                                     # :49
    lam_max = float(spla.eigsh(S, k=1, which="LA", return_eigenvectors=False, tol=1e-10)[0])
    hi, lo = 1e-4, -1e-3
    if _neg_count(S, hi) == 0:
        lam_min = hi                                                  # positive definite beyond 1e-4: dinf = 0
    else:
        while _neg_count(S, lo) > 0:
            lo *= 4.0
        while hi - lo > 1e-14:
            mid = 0.5 * (lo + hi)
            if _neg_count(S, mid) > 0:
                hi = mid
            else:
                lo = mid
        lam_min = 0.5 * (lo + hi)
    return max(0.0, -lam_min) / (1.0 + lam_max), lam_min, lam_max, z


def test_g81_kkt_certificate_is_confirmed_on_the_host(lib):
    """p0 = 40 (example_maxcut.m:32): the solve certifies dinf < 1e-8 with the device escape, and ARPACK shift-invert on
    the host confirms that number.  Default p0 = 2: the reference's algorithm crawls on this ill-conditioned instance
    (every RTR call exhausts its 40 x 100 budget from the fourth outer iteration on, the factor settles at r = 18,
    p = 26 and dinf levels off near 8e-8 until the reference's own "Slow progress" exit -- SURVEY.md section 6 saw the same
    with SciPy), so for that run the test pins what must hold regardless: the reported dinf is the true one and the
    objective is the optimum to the accuracy that dinf implies."""
    from manisdp_matlab_amd import problems, solvers
    C = problems.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
    assert data["status"] == 0 and data["dinf"] < 1e-8, (data["status"], data["dinf"], data["iters"])
    assert data.get("eig_unconverged", 0) == 0
    dinf_host, lam_min, lam_max, z = _host_dinf(C, Y)
    assert dinf_host < 1e-8
    assert abs(dinf_host - data["dinf"]) <= 1e-9
    assert abs(obj - float(np.sum(z))) <= 1e-9 * abs(obj)
    assert np.abs(np.linalg.norm(Y, axis=1) - 1.0).max() < 1e-12      # feasible: unit diagonal of X = YY'
    Y2, obj2, d2 = solvers.ManiSDP_onlyunitdiag(C, {"p0": 2, "AL_maxiter": 24}, verbose=False)
    assert d2.get("eig_unconverged", 0) == 0
    dinf_host2, _, _, _ = _host_dinf(C, Y2)
    assert abs(dinf_host2 - d2["dinf"]) <= 1e-9                       # whatever it is, it is the true dinf
    assert d2["dinf"] < 1e-6 and (d2["status"] == 0) == (d2["dinf"] < 1e-8)
    # weak duality: obj - n*max(0, -lambda_min) <= optimum <= obj for both runs
    assert abs(obj2 - obj) <= 20000 * 5.0 * max(d2["dinf"], data["dinf"])


def test_escape_reports_unconverged_runs_and_missing_pairs(lib):
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G11.txt.gz"))       # n = 800
    n = C.shape[0]
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, 3)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    # far too few steps for 1e-12: the run must say so
    lam, V, lmax, its = h.escape_eigs(4, tol=1e-12, maxit=8)
    nvalid, conv, res = h.escape_info()
    assert not conv and res > 1e-12 and 1 <= nvalid <= 4
    # enough steps: converged, and the values are LAPACK's
    lam, V, lmax, its = h.escape_eigs(4, tol=1e-10, maxit=800)
    nvalid, conv, res = h.escape_info()
    assert conv and res == 0.0 and nvalid == 4
    z = h.get_z()
    dS = np.linalg.eigvalsh(C.toarray() - np.diag(z))
    assert np.allclose(lam, dS[:4], rtol=0, atol=1e-8 * abs(dS[0]))
    h.close()
    # a PSD slack: C = -adjacency of a ring, Y = all-ones (p = 1) gives S = C - diag(C*1) = the graph Laplacian, whose
    # kernel is span(Y).  Two real pairs exist in the searched space (0 from span(Y), lambda_2 from the one Lanczos
    # run); the other three are +inf with zero vectors and are never counted as negative.
    n2 = 300
    i = np.arange(n2)
    A = sp.csr_matrix((np.ones(2 * n2), (np.r_[i, i], np.r_[(i + 1) % n2, (i - 1) % n2])), shape=(n2, n2))
    h = lib.Handle.onlyunitdiag(sp.csr_matrix(-A))
    h.set_point(np.ones((n2, 1)))
    lam, V, lmax, its = h.escape_eigs(5, tol=1e-10, maxit=300)
    nvalid, conv, _ = h.escape_info()
    assert conv and nvalid == 2
    lam2 = 2 - 2 * np.cos(2 * np.pi / n2)                             # the run stops once lambda_2 is CERTIFIED positive
    assert abs(lam[0]) < 1e-9 and 0 < lam[1] and abs(lam[1] - lam2) < 1e-2 * lam2
    assert np.all(np.isinf(lam[nvalid:])) and np.all(lam[nvalid:] > 0)
    assert np.all(V[:, nvalid:] == 0.0)
    assert int(np.sum(lam < 0)) <= 1
    assert abs(lmax - 4.0) < 1e-3                                        # lambda_max only sets the scale of dinf
    h.close()


def test_al_loop_does_not_certify_on_an_unconverged_escape(lib):
    """With a Lanczos budget that cannot resolve lambda_min the solver must not print "Optimality is reached"."""
    from manisdp_matlab_amd import problems, solvers
    C = problems.maxcut_cost_matrix(golden_path("G11.txt.gz"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": "device", "eig_maxit": 8, "eig_tol": 1e-13, "AL_maxiter": 6},
                                                verbose=False)
    assert data["status"] == 1                                        # iteration limit, not converged
    assert data.get("eig_unconverged", 0) >= 1
    # the same instance with the default budget certifies
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": "device"}, verbose=False)
    assert data["status"] == 0 and data["dinf"] < 1e-8 and data.get("eig_verifications", 0) >= 1


# ------------------------------------------------------------------------------------------------ a13
@pytest.mark.parametrize("p,q", [(3, 2), (12, 8)])
def test_onlyunitdiag_line_search_operators(lib, p, q):
    """co(Y) = sum((Y*C).*Y) (ManiSDP_onlyunitdiag.m:99-101) at the retracted trial points of line_search (:103-115),
    with the escape-direction shape U = [0; vS'] (:75-77)."""
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = np.hstack([rng.standard_normal((n, p)), np.zeros((n, q))])
    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = np.hstack([np.zeros((n, p)), np.linalg.qr(rng.standard_normal((n, q)))[0]])

    def co(Z):
        return float(np.sum((C @ Z) * Z))

    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    assert abs(h.linesearch_cost(None, 0.0) - co(Y)) <= 1e-12 * abs(co(Y))
    for alpha in (1.0, 0.8, 0.8 ** 7):
        Z = Y + alpha * U
        Z /= np.linalg.norm(Z, axis=1, keepdims=True)
        assert abs(h.linesearch_cost(U, alpha) - co(Z)) <= 1e-12 * abs(co(Z))
    # accepting the last trial makes it the resident point
    h.linesearch_accept()
    assert np.linalg.norm(h.get_point() - Z) <= 1e-14 * np.linalg.norm(Z)
    h.close()


@pytest.mark.parametrize("graph", ["G1", "G11"])
def test_onlyunitdiag_solver_with_line_search_matches_oracle(lib, graph):
    from manisdp_matlab_amd import problems, solvers
    from oracle import manisdp_ref as R
    C = problems.maxcut_cost_matrix(golden_path(graph + ".txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(1)
    Y0 = rng.standard_normal((n, 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    Yr, objr, dr = R.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "line_search": 1}, q1="correct")
    assert dr["status"] == 0 and dr["dinf"] < 1e-8
    for mode in ("host", "device"):
        Y, obj, d = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "line_search": 1, "eig": mode}, verbose=False)
        assert d["status"] == 0 and d["dinf"] < 1e-8
        assert abs(obj - objr) <= 1e-6 * abs(objr)
