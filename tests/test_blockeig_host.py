"""Host-side dense algebra of the block eigen-solver (manisdp-matlab_amd/csrc/msdp_blockeig.hip) against LAPACK.
CPU only: the two test entry points run no device code."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

from manisdp_matlab_amd import _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 128])
def test_sym_eig_matches_lapack(n):
    lib = _lib.load()
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    A = A + A.T
    if n >= 17:                                       # a cluster and a few exact repeats, like the bottom of S
        A = A @ np.diag(np.r_[np.zeros(5), 1e-9 * rng.standard_normal(4), rng.standard_normal(n - 9)]) @ A.T
        A = 0.5 * (A + A.T)
    w = np.empty(n)
    Z = np.empty((n, n))
    assert lib.msdp_debug_sym_eig(n, _dp(np.ascontiguousarray(A)), _dp(w), _dp(Z)) == 0
    wl = np.linalg.eigvalsh(A)
    scale = max(1.0, np.abs(wl).max())
    assert np.all(np.diff(w) >= 0)
    assert np.abs(w - wl).max() <= 1e-13 * scale * n
    # rows of Z are orthonormal eigenvectors
    assert np.abs(Z @ Z.T - np.eye(n)).max() <= 1e-12 * n
    assert np.abs(A @ Z.T - Z.T * w).max() <= 1e-12 * scale * n


@pytest.mark.parametrize("b,cond", [(32, 1e2), (64, 1e6), (128, 1e3)])
def test_ritz_matches_generalised_eigh(b, cond):
    lib = _lib.load()
    rng = np.random.default_rng(b)
    n = 4 * b
    S = rng.standard_normal((n, n)); S = S + S.T
    X = rng.standard_normal((n, b)) @ np.diag(np.logspace(0, np.log10(cond) / 2, b))
    G = X.T @ X
    H = X.T @ S @ X
    th = np.empty(b); W = np.empty((b, b)); r = C.c_int32()
    assert lib.msdp_debug_ritz(b, _dp(np.ascontiguousarray(G)), _dp(np.ascontiguousarray(H)), _dp(th), _dp(W), C.byref(r)) == 0
    assert r.value == b
    wl = sla.eigh(0.5 * (H + H.T), 0.5 * (G + G.T), eigvals_only=True)
    assert np.abs(th - wl).max() <= 1e-9 * np.abs(wl).max()
    assert np.abs(W.T @ G @ W - np.eye(b)).max() <= 1e-8
    Xr = X @ W                                         # Ritz vectors: orthonormal, S-orthogonal
    assert np.abs(Xr.T @ S @ Xr - np.diag(th)).max() <= 1e-8 * np.abs(wl).max()


def test_ritz_drops_dependent_columns():
    """A start block whose warm-start columns lie in the span of the factor columns: the Gram matrix is singular and the
    Cholesky route gives way to the eigen-basis of G; the dropped directions come back as +inf / zero columns."""
    lib = _lib.load()
    rng = np.random.default_rng(7)
    b, n = 32, 200
    S = rng.standard_normal((n, n)); S = S + S.T
    X = rng.standard_normal((n, b))
    X[:, 20:24] = X[:, :4] @ rng.standard_normal((4, 4))          # four dependent columns
    G = X.T @ X
    H = X.T @ S @ X
    th = np.empty(b); W = np.empty((b, b)); r = C.c_int32()
    assert lib.msdp_debug_ritz(b, _dp(np.ascontiguousarray(G)), _dp(np.ascontiguousarray(H)), _dp(th), _dp(W), C.byref(r)) == 0
    assert r.value == b - 4
    assert np.all(np.isinf(th[b - 4:])) and np.all(W[:, b - 4:] == 0.0)
    Q, _ = np.linalg.qr(X[:, list(range(20)) + list(range(24, b))])
    wl = np.linalg.eigvalsh(Q.T @ S @ Q)
    assert np.abs(th[:b - 4] - wl).max() <= 1e-9 * np.abs(wl).max()
    Xr = X @ W[:, :b - 4]
    assert np.abs(Xr.T @ Xr - np.eye(b - 4)).max() <= 1e-9
