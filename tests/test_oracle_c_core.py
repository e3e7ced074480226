"""CPU test: the plain-C restatement (oracle/oracle_core.c, OpenMP) and the NumPy restatement
(oracle/manopt_rtr.py + manisdp_ref.py) of the onlyunitdiag hot path pin each other."""
import numpy as np

from conftest import golden_path
from manisdp_matlab_amd import problems
from oracle import core, manisdp_ref as R, manopt_rtr


def test_c_operators_match_numpy():
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(0)
    for p in (1, 2, 7, 32):
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        U = rng.standard_normal((n, p))
        prob = R._OnlyUnitDiagProblem(C, n, p)
        f = prob.cost(Y); G = prob.grad(Y); H = prob.hess(Y, U)
        fc, eG, Gc = core.cost_state(C, Y)
        assert abs(fc - f) < 1e-12 * abs(f)
        assert np.allclose(Gc, G, rtol=0, atol=1e-12)
        assert np.allclose(core.hessvec(C, Y, U, eG), H, rtol=0, atol=1e-12)


def test_c_rtr_matches_numpy_rtr():
    C = problems.toroidal_grid_maxcut(20, 30, seed=1)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(2)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    for maxiter, maxinner in [(1, 5), (3, 20), (40, 100)]:
        prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), maxiter, maxinner, 1e-8)
        Yc, st = core.rtr_onlyunitdiag(C, Y, maxiter, maxinner, 1e-8)
        if maxiter <= 3:                          # before summation-order noise can change a decision
            assert st.hessvecs == info.hessvecs and st.iters == info.iters
            assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-9 * max(1.0, abs(f_ref))
        assert np.allclose(np.linalg.norm(Yc, axis=1), 1.0, atol=1e-13)
