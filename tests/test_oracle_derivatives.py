"""CPU tests: finite-difference checks of the oracle's gradient and Hessian closures, the procedure of
manopt/tools/checkgradient.m / checkhessian.m: no golden vectors for individual operators
exist in the reference, so the restated closures are checked against their own cost function."""
import numpy as np
import pytest

from conftest import golden_path
from manisdp_matlab_amd import problems
from oracle import manisdp_ref as R


def _fd_check(prob, Y, U, retr):
    """f(R(tU)) = f + t<G,U> + t^2/2 <U,Hess U> + O(t^3) on the manifold."""
    f0 = prob.cost(Y)
    G = prob.grad(Y)
    H = prob.hess(Y, U)
    g1 = float(np.sum(G * U))
    g2 = float(np.sum(U * H))
    errs = []
    for t in (1e-2, 5e-3, 2.5e-3):
        ft = prob.cost(retr(Y, t * U))
        errs.append(abs(ft - (f0 + t * g1 + 0.5 * t * t * g2)))
    prob.cost(Y); prob.grad(Y)          # restore shared state
    # third-order remainder: halving t divides the error by ~8
    assert errs[0] / max(errs[1], 1e-300) > 5.0 and errs[1] / max(errs[2], 1e-300) > 5.0, errs


def test_fd_onlyunitdiag():
    C = problems.toroidal_grid_maxcut(12, 15, seed=2)
    n, p = C.shape[0], 5
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    prob = R._OnlyUnitDiagProblem(C, n, p)
    U = prob.M.proj(Y, rng.standard_normal((n, p)))
    _fd_check(prob, Y, U, prob.M.retr)


def test_fd_unitdiag():
    At, b, c, K = problems.from_sdpa(golden_path("gpp100.dat-s.gz"))
    c = np.asarray(c.todense()).ravel()
    n, p = K["s"], 4
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    prob = R._UnitDiagProblem(At, np.asarray(b, float), c, n, p)
    prob.y = 0.1 * rng.standard_normal(b.size); prob.sigma = 0.5
    U = prob.M.proj(Y, rng.standard_normal((n, p)))
    _fd_check(prob, Y, U, prob.M.retr)


def test_fd_unittrace():
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    c = np.asarray(c.todense()).ravel()
    n, p = K["s"], 3
    rng = np.random.default_rng(2)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
    prob = R._UnitTraceProblem(At, np.asarray(b, float), c, n, p)
    prob.y = 0.1 * rng.standard_normal(b.size); prob.sigma = 3.0
    U = prob.M.proj(Y, rng.standard_normal((n, p)))
    _fd_check(prob, Y, U, prob.M.retr)


def test_tcg_stop_codes_and_trust_region():
    """tCG exits: with a tiny radius the first step hits the boundary (stop 2), |eta| = Delta."""
    from oracle import manopt_rtr
    C = problems.toroidal_grid_maxcut(10, 10, seed=5)
    n, p = C.shape[0], 3
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    prob = R._OnlyUnitDiagProblem(C, n, p)
    prob.cost(Y); g = prob.grad(Y)
    eta, Heta, it, stop = manopt_rtr.tCG(prob, Y, g, 1e-3, 50)
    assert stop in (1, 2) and it == 1
    assert abs(np.linalg.norm(eta) - 1e-3) < 1e-12
    eta, Heta, it, stop = manopt_rtr.tCG(prob, Y, g, 1e3, 1)
    assert stop == 5 or stop in (1, 2, 3, 4, 6)
