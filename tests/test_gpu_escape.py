"""GPU tests of the few-eigenvector saddle escape (device Lanczos with deflation of span(Y)) against the
reference's dense eig(full(S)) (ManiSDP_onlyunitdiag.m:49-51) computed with LAPACK on the same S."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


def _check(lib, C, Y, k=8, rtr=True):
    import scipy.sparse as sp
    n = C.shape[0]
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    if rtr:
        h.rtr(lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
    Yc = h.get_point()
    z = h.get_z()
    Cd = C.toarray() if sp.issparse(C) else C
    S = Cd - np.diag(z)
    dS, vS = np.linalg.eigh(S)
    lam, V, lmax, its = h.escape_eigs(k, tol=1e-9, maxit=600)
    h.close()
    scale = max(abs(dS[0]), abs(dS[-1]))
    nneg = int(np.sum(dS < -1e-9 * scale))
    # lambda_max and lambda_min to Ritz accuracy
    assert abs(lmax - dS[-1]) < 1e-5 * scale
    assert abs(lam[0] - dS[0]) < 1e-8 * scale
    # every returned negative eigenvalue is a true eigenvalue with a small residual
    for t in range(k):
        if lam[t] < -1e-9 * scale:
            v = V[:, t]
            assert abs(np.linalg.norm(v) - 1.0) < 1e-8
            assert np.linalg.norm(S @ v - lam[t] * v) < 1e-5 * scale     # escape directions; looser when S*Y != 0
            assert np.min(np.abs(dS - lam[t])) < 1e-8 * scale
    # dinf as the AL loop computes it
    dinf_ref = max(0.0, -dS[0]) / (1 + dS[-1])
    dinf = max(0.0, -lam[0]) / (1 + lmax)
    assert abs(dinf - dinf_ref) < 1e-8 * max(1.0, dinf_ref)
    return nneg, lam


def test_escape_at_stationary_point_G1(lib):
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    rng = np.random.default_rng(0)
    for p in (2, 10):
        Y = rng.standard_normal((C.shape[0], p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        nneg, lam = _check(lib, C, Y)
        assert nneg > 0 and lam[0] < 0        # rank-p stationary points of G1 are saddles for small p


def test_escape_random_point_and_dense(lib):
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(20, 25, seed=3)
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((C.shape[0], 6)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    _check(lib, C, Y, rtr=False)               # S*Y != 0: the final Rayleigh-Ritz recouples the blocks
    _check(lib, C.toarray(), Y, rtr=True)      # dense-C S*v kernel


def test_escape_at_optimum_certifies_psd(lib):
    """At the SDP optimum S is PSD with a kernel of dimension rank(Y): lambda_min ~ 0, dinf < 1e-8."""
    from manisdp_matlab_amd import problems, solvers
    C = problems.maxcut_cost_matrix(golden_path("G11.txt.gz"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {}, verbose=False)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    lam, V, lmax, its = h.escape_eigs(8, tol=1e-9, maxit=800)
    h.close()
    assert max(0.0, -lam[0]) / (1 + lmax) < 1e-7


def test_solver_with_device_escape(lib):
    """Full AL loop with the device escape instead of host eig: same optimum (SDPLIB maxG11)."""
    import json
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    C = problems.maxcut_cost_matrix(golden_path("G11.txt.gz"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": "device"}, verbose=False)
    assert data["dinf"] < 1e-8
    assert abs(-obj - known["maxG11"]) < 1e-6 * known["maxG11"]


def test_onesync_lanczos_matches_twosync_kernel(lib):
    """Undeflated persistent Lanczos runs: the one-synchronisation kernel (beta from |w'|^2 - alpha^2 |v|^2, neighbour
    entries rebuilt from the published pair) against the two-synchronisation kernel on the same S (toroidal grid, n = 6000)
    and against LAPACK."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(60, 100, seed=9)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = []
    for one in (1, 0):
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("escape_method", 1)                      # the Lanczos path (n = 6000 would take the block eigen-solver)
        h.set_option("lanczos_onesync", one)
        h.set_option("escape_deflate", 0)
        h.set_option("escape_warm", 0)
        h.set_point(Y)
        lam, V, lmax, steps = h.escape_eigs(1, tol=1e-10, maxit=20000)
        _, conv, _ = h.escape_info()
        z = h.get_z()
        out.append((lam[0], lmax, V[:, 0].copy(), conv))
        h.close()
    S = (C - __import__("scipy.sparse", fromlist=["diags"]).diags(z)).toarray()
    w = np.linalg.eigvalsh(S)
    for lam0, lmax, v, conv in out:
        assert conv
        assert abs(lam0 - w[0]) <= 1e-8 * max(1.0, abs(w[-1]))
        assert abs(lmax - w[-1]) <= 1e-6 * abs(w[-1])
        assert np.linalg.norm(S @ v - lam0 * v) <= 1e-6 * abs(w[-1])
    assert abs(out[0][0] - out[1][0]) <= 1e-9 * abs(w[-1])


def test_deflation_columns_read_in_place_are_bit_identical(lib):
    """Deflated persistent Lanczos runs with the deflation columns read from Q itself (what n > ~117 000 falls back to when the
    LDS copy does not fit; option lanczos_qglobal forces it on the same grid) against the LDS copy: same arithmetic, same bits.
    Four pairs at a random point (every accepted vector joins the deflation set) and the span(Y)-deflated run at a
    stationary point."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(60, 100, seed=9)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = []
    for inplace in (0, 1):
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("escape_method", 1)
        h.set_option("lanczos_qglobal", inplace)
        h.set_option("escape_warm", 0)
        h.set_point(Y)
        lam, V, lmax, steps = h.escape_eigs(4, tol=1e-10, maxit=20000)
        for _ in range(12):                                  # near-stationary: span(Y) is deflated (|grad| <= 1e-6 |f|)
            st = h.rtr(lib.default_opts(maxiter=100, maxinner=400, tolgradnorm=1e-9))
            if st.gradnorm < 1e-3:
                break
        lam2, V2, lmax2, steps2 = h.escape_eigs(2, tol=1e-10, maxit=20000)
        out.append((lam, V, lmax, steps, lam2, V2, lmax2, steps2, st.gradnorm))
        h.close()
    a, b = out
    assert a[8] < 1e-3                                       # the second call did deflate span(Y)
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
