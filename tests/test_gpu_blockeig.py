"""GPU tests of the block eigen-solver of the saddle escape (msdp_blockeig.hip: Chebyshev-filtered subspace iteration on a
64- / 128-wide panel) against the reference's dense eig(full(S)) (ManiSDP_onlyunitdiag.m:49-51) computed with LAPACK on the
same S = C - diag(z), through the C ABI (msdp_escape_eigs)."""
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


def _reference(C, h):
    z = h.get_z()
    S = (C - sp.diags(z)).toarray()
    w, U = np.linalg.eigh(S)
    return S, w, U


def _check_pairs(S, w, lam, V, lmax, k, tol_val, tol_res):
    scale = max(abs(w[0]), abs(w[-1]))
    assert abs(lmax - w[-1]) <= 1e-6 * scale
    assert abs(lam[0] - w[0]) <= tol_val * scale
    for t in range(k):
        if lam[t] < -1e-9 * scale or t == 0:
            v = V[:, t]
            assert abs(np.linalg.norm(v) - 1.0) < 1e-8
            assert np.min(np.abs(w - lam[t])) <= 10 * tol_val * scale
            assert np.linalg.norm(S @ v - lam[t] * v) <= tol_res * scale
    # the returned values are the BOTTOM of the spectrum, in order
    assert np.all(np.diff(lam[np.isfinite(lam)]) >= -1e-12 * scale)
    nneg = int(np.sum(w < -1e-7 * scale))
    assert int(np.sum(lam < -1e-7 * scale)) == min(nneg, k)


@pytest.mark.parametrize("p", [6, 30, 60])
def test_block_escape_matches_lapack_on_a_grid(lib, p):
    """Toroidal grid (ELL rows, the G81 family), n = 6000: a random point (S*Y != 0), then a near-stationary one; warm
    calls and the cold-started check; p = 60 takes the 128-wide panel."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(60, 100, seed=9)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = lib.Handle.onlyunitdiag(C, pcap=64)
    h.set_point(Y)
    S, w, _ = _reference(C, h)
    lam, V, lmax, deg = h.escape_eigs(8, tol=1e-9, maxit=60000)
    assert h.escape_method() == 1
    nvalid, conv, _ = h.escape_info()
    assert conv and nvalid == 8
    assert h.escape_lower_bound() == -np.inf                   # a warm call never reports a bound
    _check_pairs(S, w, lam, V, lmax, 8, 2e-3, 2e-2)            # escape directions: 2 % of |theta| asked
    h.rtr(lib.default_opts(maxiter=60, maxinner=200, tolgradnorm=1e-9))
    S, w, _ = _reference(C, h)
    lam, V, lmax, deg2 = h.escape_eigs(8, tol=1e-9, maxit=60000)
    _, conv, _ = h.escape_info()
    assert conv
    _check_pairs(S, w, lam, V, lmax, 8, 2e-3, 2e-2)
    # the independent check: cold start, nothing of Y, four times tighter
    h.set_option("escape_deflate", 0); h.set_option("escape_warm", 0); h.set_option("escape_start_y", 0)
    lam1, V1, lmax1, deg3 = h.escape_eigs(1, tol=1e-9, maxit=60000)
    _, conv, _ = h.escape_info()
    assert conv
    scale = max(abs(w[0]), abs(w[-1]))
    assert abs(lam1[0] - w[0]) <= 1e-9 * scale
    assert np.linalg.norm(S @ V1[:, 0] - lam1[0] * V1[:, 0]) <= 1e-5 * scale
    lb = h.escape_lower_bound()
    assert np.isfinite(lb) and lb <= lam1[0] and lam1[0] - lb <= 1e-8 * scale
    h.close()


def test_block_escape_long_rows_and_small_n(lib):
    """G1 (n = 800, ~48 entries per row: the CSR gather, no ELL copy), forced onto the block path."""
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(0)
    for p in (2, 10):
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("escape_method", 2)
        h.set_point(Y)
        h.rtr(lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        S, w, _ = _reference(C, h)
        lam, V, lmax, deg = h.escape_eigs(8, tol=1e-9, maxit=60000)
        assert h.escape_method() == 1
        _, conv, _ = h.escape_info()
        assert conv
        _check_pairs(S, w, lam, V, lmax, 8, 2e-3, 2e-2)
        assert lam[0] < 0                                       # rank-p stationary points of G1 are saddles for small p
        h.close()


def test_block_and_lanczos_paths_agree(lib):
    """The two eigen-solvers on the same S at a near-stationary point: same lambda_min / lambda_max / dinf."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(50, 80, seed=2)
    n, p = C.shape[0], 12
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = []
    for method in (1, 2):
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("escape_method", method)
        h.set_point(Y)
        h.rtr(lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        h.set_option("escape_deflate", 0); h.set_option("escape_warm", 0)
        lam, V, lmax, _ = h.escape_eigs(1, tol=1e-10, maxit=60000)
        assert h.escape_method() == method - 1
        out.append((lam[0], lmax))
        h.close()
    assert abs(out[0][0] - out[1][0]) <= 1e-8 * out[0][1]
    assert abs(out[0][1] - out[1][1]) <= 1e-6 * out[0][1]


def test_block_escape_budget_exhausted_is_reported(lib):
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(60, 100, seed=9)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    h.set_option("escape_method", 2)                              # (0 would repeat an unconverged block call on the Lanczos path)
    h.set_option("escape_deflate", 0); h.set_option("escape_warm", 0)
    lam, V, lmax, deg = h.escape_eigs(4, tol=1e-13, maxit=64)     # 64 filter steps cannot reach 1e-13
    nvalid, conv, res = h.escape_info()
    assert not conv and res > 1e-13 and nvalid == 4
    assert h.escape_lower_bound() == -np.inf
    h.close()


@pytest.mark.parametrize("kind", ["dense_C", "explicit_S"])
def test_block_escape_on_a_dense_operand_matches_lapack(lib, kind):
    """The dense route of the block eigen-solver (filter step = fp64-MFMA panel product + epilogue): S = C - diag(z) with a dense C
    (onlyunitdiag, n = 1500) and an explicit dense S handed over by the AL loop of the affine kinds (msdp_escape_eigs_matrix),
    against LAPACK on the same matrix; warm call and the cold-started check."""
    from manisdp_matlab_amd import problems
    rng = np.random.default_rng(3)
    if kind == "dense_C":
        n, p = 1500, 12
        C = problems.dense_unitdiag_cost(n, seed=1)
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_point(Y)
        h.rtr(lib.default_opts(maxiter=30, maxinner=60, tolgradnorm=1e-9))
        S = C - np.diag(h.get_z())
        run = lambda k: h.escape_eigs(k, tol=1e-9, maxit=60000)
    else:
        At, b, c, K = problems.theta_problem(1200, ndraws=6000, seed=2)
        c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
        n, p = K["s"], 6
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
        h = lib.Handle.affine(lib.KIND_UNITTRACE, At, np.asarray(b, float), c, n, pcap=p)
        h.set_option("escape_method", 2)                       # explicit S: the Lanczos path unless asked
        h.set_multipliers(np.zeros(len(b)), 10.0)
        h.set_point(Y)
        h.rtr(lib.default_opts(maxiter=10, maxinner=40, tolgradnorm=1e-9))
        G = rng.standard_normal((n, n)); S = (G + G.T) / np.sqrt(n)
        S[:40, :40] += -0.5 * np.eye(40)                       # a few well separated negative directions
        S = 0.5 * (S + S.T)
        run = lambda k: h.escape_eigs_matrix(S, k, tol=1e-9, maxit=60000)
    w = np.linalg.eigvalsh(S)
    scale = max(abs(w[0]), abs(w[-1]))
    lam, V, lmax, deg = run(8)
    assert h.escape_method() == 1                               # n >= 1024: the block path
    nvalid, conv, _ = h.escape_info()
    assert conv and nvalid == 8
    assert abs(lmax - w[-1]) <= 1e-6 * scale
    assert np.abs(lam - w[:8]).max() <= 2e-3 * scale and abs(lam[0] - w[0]) <= 1e-6 * scale
    for t in range(8):
        assert np.linalg.norm(S @ V[:, t] - lam[t] * V[:, t]) <= 2e-2 * scale
    h.set_option("escape_deflate", 0); h.set_option("escape_warm", 0); h.set_option("escape_start_y", 0)
    lam1, V1, lmax1, _ = run(1)
    _, conv, _ = h.escape_info()
    assert conv and abs(lam1[0] - w[0]) <= 1e-8 * scale
    assert np.linalg.norm(S @ V1[:, 0] - lam1[0] * V1[:, 0]) <= 1e-4 * scale
    h.close()


def test_block_escape_lengthens_its_rounds_on_a_dense_bottom_cluster(lib):
    """The G81 family at n = 80 000 (toroidal 400 x 200 grid; bench.py --gpus 4 solves it sharded): near its optimum the bottom
    of the spectrum of S is a cluster of more than 64 eigenvalues within a few 1e-6 of the width.  Rounds of 200 filter steps
    gain cosh(200 acosh x0) ~ 1.1 there and the calls used to stall (the Lanczos path took over: 12-57 k steps each at
    n = 160 000); with the rounds lengthened to cosh(3) every call converges on the block path.  Whole solve: KKT 1e-8, no
    escape call needed a second attempt or ended unconverged."""
    from manisdp_matlab_amd import problems, solvers
    C = problems.toroidal_grid_maxcut(400, 200, seed=81)
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert data.get("eig_retries", 0) == 0 and data.get("eig_unconverged", 0) == 0
    assert data["escape_method"] == 1                      # the last regular escape call ran the block eigen-solver
    assert abs(obj + 62417.9192527) <= 1e-6 * 62417.9      # the value both rounds' builds and the Lanczos escape reach


def test_unconverged_block_call_is_repeated_on_the_lanczos_path(lib):
    """msdp_escape_eigs never hands an unconverged block result to the host loops when the Lanczos path can do better: a block
    call that ends without passing its stop test (forced here by the test hook debug_fail_block; in the field: a bottom cluster
    wider than the panel with no gap behind it) is repeated on the Lanczos path inside the same C call.  The outcome is the
    Lanczos run's: converged, lambda_min right against LAPACK, msdp_escape_method = 0, both attempts counted in the steps; with
    escape_method = 2 (block only) the same call reports the failure instead."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(50, 60, seed=9)            # n = 3000: the block eigen-solver's default territory
    n, p = C.shape[0], 6
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = {}
    for method in (0, 2):
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("escape_method", method)
        h.set_option("escape_deflate", 0); h.set_option("escape_warm", 0)
        h.set_point(Y)
        lam_b, _, _, steps_b = h.escape_eigs(1, tol=1e-10, maxit=2000)       # the block path, undisturbed
        assert h.escape_method() == 1 and h.escape_info()[1]
        h.set_option("debug_fail_block", 1)
        lam, V, lmax, steps = h.escape_eigs(1, tol=1e-10, maxit=2000)
        nvalid, conv, res = h.escape_info()
        out[method] = (lam[0], V[:, 0].copy(), lmax, conv, h.escape_method(), steps, steps_b, lam_b[0])
        if method == 0:
            S, w, U = _reference(C, h)
        h.close()
    lam0, v, lmax, conv, ran, steps, steps_b, lam_b = out[0]
    assert conv and ran == 0 and steps > steps_b                 # the Lanczos run's outcome; both attempts are counted
    scale = max(abs(w[0]), abs(w[-1]))
    assert abs(lam0 - w[0]) <= 1e-8 * scale and abs(lam_b - w[0]) <= 1e-8 * scale and abs(lmax - w[-1]) <= 1e-6 * scale
    assert np.linalg.norm(S @ v - lam0 * v) <= 1e-6 * scale
    assert not out[2][3] and out[2][4] == 1                      # block only: unconverged, and said so


def test_factor_wider_than_the_panel_takes_the_lanczos_path(lib):
    """The panel must hold span(Y) and noise columns beside it (msdp_blockeig_eligible): a factor of more than 112 columns takes
    the Lanczos path (sparse C, 128-wide panel); 100 columns still run on the block path.  Same lambda_min either way."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(50, 50, seed=2)            # n = 2500
    n = C.shape[0]
    rng = np.random.default_rng(1)
    vals = {}
    for p in (100, 120):
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_point(Y)
        lam, V, lmax, steps = h.escape_eigs(2, tol=1e-9, maxit=20000)
        _, conv, _ = h.escape_info()
        S, w, U = _reference(C, h)
        assert conv and h.escape_method() == (1 if p == 100 else 0)
        assert abs(lam[0] - w[0]) <= 1e-8 * max(abs(w[0]), abs(w[-1]))
        h.close()
