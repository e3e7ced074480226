"""GPU tests: finite-difference checks of cost, Riemannian gradient and Hess-vec of every kind THROUGH THE C ABI
(msdp_cost / msdp_rgrad / msdp_hessvec / msdp_proj / msdp_retr), the procedure of manopt/tools/checkgradient.m and
checkhessian.m: on the manifold f(R_Y(tU)) = f + t<G,U> + t^2/2 <U, Hess U> + O(t^3), so halving t divides the residual of
the second-order model by ~8.  Independent of the oracle: the device closures are checked against the device cost alone
(the oracle passes the same check on the CPU, tests/test_oracle_derivatives.py)."""
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


def _fd_check(h, Y, Uraw):
    h.set_point(Y)
    f0 = h.cost()
    G = h.rgrad()
    U = h.proj(Uraw)
    U *= np.linalg.norm(Y) / np.linalg.norm(U)             # a step t*U moves every point by ~t times its own size
    H = h.hessvec(U)
    g1 = float(np.sum(G * U))
    g2 = float(np.sum(U * H))
    # the gradient is tangent, the Hess-vec too (ManiSDP_onlyunitdiag.m:129 projects; spherefactory.m:229)
    assert np.linalg.norm(h.proj(G) - G) <= 1e-12 * max(1.0, np.linalg.norm(G))
    assert np.linalg.norm(h.proj(H) - H) <= 1e-11 * max(1.0, np.linalg.norm(H))
    errs, first = [], []
    for t in (2e-2, 1e-2, 5e-3):
        Z = h.retr(t * U)                                  # retraction from the resident point, which stays put
        h.set_point(Z)
        ft = h.cost()
        h.set_point(Y)
        h.cost()                                           # per-point state back at Y for the next retraction
        errs.append(abs(ft - (f0 + t * g1 + 0.5 * t * t * g2)))
        first.append(abs(ft - (f0 + t * g1)))
    # first-order model: error O(t^2) (ratio ~4); second-order model: O(t^3) (ratio ~8)
    assert 3.0 < first[0] / first[1] < 5.5 and 3.0 < first[1] / first[2] < 5.5, first
    assert errs[0] / max(errs[1], 1e-300) > 5.5 and errs[1] / max(errs[2], 1e-300) > 5.5, errs
    # symmetry of the Hessian on the tangent space: <V, Hess U> = <U, Hess V>
    V = h.proj(np.random.default_rng(99).standard_normal(U.shape))
    HV = h.hessvec(V)
    a, b = float(np.sum(V * H)), float(np.sum(U * HV))
    assert abs(a - b) <= 1e-10 * max(1.0, abs(a), abs(b))


@pytest.mark.parametrize("p", [3, 20, 70])
def test_fd_onlyunitdiag_sparse_and_dense(lib, p):
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(30, 40, seed=2)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    for Cm in (C, C.toarray()):
        h = lib.Handle.onlyunitdiag(Cm, pcap=p)
        _fd_check(h, Y, U)
        h.close()


def _affine(lib, kind, name, p, sigma, seed):
    from manisdp_matlab_amd import problems
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
    n = K["s"]
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, p))
    if kind == lib.KIND_UNITDIAG:
        Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    elif kind == lib.KIND_UNITTRACE:
        Y /= np.linalg.norm(Y)
    h = lib.Handle.affine(kind, At, b, c, n, pcap=p)
    h.set_multipliers(0.1 * rng.standard_normal(b.size), sigma)
    return h, Y, rng.standard_normal((n, p))


@pytest.mark.parametrize("route", [1, 2])
def test_fd_unitdiag(lib, route):
    h, Y, U = _affine(lib, lib.KIND_UNITDIAG, "gpp100", 4, 0.5, 1)
    h.set_option("affine_route", route)                    # SDDMM and Gram route of A(Ya Yb')
    _fd_check(h, Y, U)
    h.close()


@pytest.mark.parametrize("name,p", [("theta1", 3), ("theta2", 12)])
def test_fd_unittrace(lib, name, p):
    h, Y, U = _affine(lib, lib.KIND_UNITTRACE, name, p, 3.0, 2)
    _fd_check(h, Y, U)
    h.close()


def test_fd_generic(lib):
    h, Y, U = _affine(lib, lib.KIND_GENERIC, "mcp100", 5, 0.7, 3)
    _fd_check(h, 0.3 * Y, U)
    h.close()
