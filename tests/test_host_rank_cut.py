"""CPU test of the host side of the rank step (ManiSDP_onlyunitdiag.m:52-54,70-73; ManiSDP_unitdiag.m:88-92,110-113): the
product path decides the rank from the p x p Gram matrix and cuts with Y*Q(:,1:r) instead of forming the n x n V of
svd(Y); both must agree with the reference formula  [~,D,V] = svd(Y); e = diag(D); r = sum(e >= theta*e(1));
Y = V(:,1:r)'.*e(1:r)  -- same r, same X = Y'Y after the cut."""
import numpy as np
import pytest

from manisdp_matlab_amd import solvers


@pytest.mark.parametrize("n,p,true_rank,theta", [(60, 7, 3, 1e-1), (200, 12, 12, 1e-3), (40, 5, 1, 1e-2)])
def test_gram_rank_decision_and_cut_match_the_svd_formula(n, p, true_rank, theta):
    rng = np.random.default_rng(n + p)
    Y = rng.standard_normal((n, true_rank)) @ rng.standard_normal((true_rank, p)) + 1e-6 * rng.standard_normal((n, p))
    # reference formula on the MATLAB layout (p x n): svd(Y') -> V is n x n, only its first r columns are used
    Uu, e_ref, Vt = np.linalg.svd(Y.T, full_matrices=False)
    r_ref = int(np.sum(e_ref >= theta * e_ref[0]))
    Ycut_ref = (Vt[:r_ref].T * e_ref[:r_ref])                     # (V(:,1:r)'.*e(1:r))' as an n x r factor
    Q, e, r = solvers._thin_svd_rank(Y, theta)
    assert r == r_ref == true_rank
    assert np.allclose(e[:r], e_ref[:r], rtol=1e-10)
    Ycut = solvers._rank_cut(Y, Q, e, r)
    assert Ycut.shape == (n, r)
    assert np.allclose(Ycut @ Ycut.T, Ycut_ref @ Ycut_ref.T, atol=1e-9 * e_ref[0] ** 2)     # same X, factor up to column signs
    # the device hands over the Gram matrix instead of Y (msdp_factor_gram): same decision
    Q2, e2, r2 = solvers._thin_svd_rank(None, theta, gram=Y.T @ Y)
    assert r2 == r and np.allclose(e2, e) and np.allclose(np.abs(Q2), np.abs(Q))


def test_rank_threshold_is_inclusive_in_the_primal_files_and_strict_in_the_dual():
    """sum(e >= theta*e(1)) (ManiSDP_onlyunitdiag.m:54) against sum(e > theta*e(1)) (ManiDSDP_unitdiag.m:90): a singular
    value exactly at the threshold counts in the primal entry points only."""
    from oracle import manisdp_ref as R
    Y = np.zeros((6, 3)); Y[0, 0] = 2.0; Y[1, 1] = 1.0; Y[2, 2] = 0.25
    _, e, r = solvers._thin_svd_rank(Y, 0.5)
    assert np.allclose(e, [2.0, 1.0, 0.25]) and r == 2
    _, e_s, r_s = R._thin_svd_rank_strict(Y, 0.5)
    assert np.allclose(e_s, [2.0, 1.0, 0.25]) and r_s == 1


class _FakeHandle:
    """Stands in for the device in solvers._line_search: co(alpha) is a given function."""
    def __init__(self, co):
        self.co, self.calls, self.accepted = co, [], 0

    def linesearch_cost(self, U, alpha):
        self.calls.append(alpha)
        return self.co(alpha)

    def linesearch_accept(self):
        self.accepted += 1


def test_line_search_backtracking_schedule():
    """line_search (ManiSDP_onlyunitdiag.m:103-115): alpha = 1, then x 0.8 at most 15 times until
    co(nY) - co(Y) <= -1e-3; the last trial point is accepted either way."""
    # decrease only for alpha <= 0.5: 1, .8, .64, .512, .4096 -> accepted at the fifth trial
    h = _FakeHandle(lambda a: 0.0 if a == 0.0 else (-1.0 if a <= 0.5 else +1.0))
    solvers._line_search(h, object())
    assert h.accepted == 1 and h.calls[0] == 0.0
    assert np.allclose(h.calls[1:], [1.0, 0.8, 0.64, 0.512, 0.4096])
    # a decrease smaller than 1e-3 does not count; after 15 reductions the search stops (16 trials in all)
    h = _FakeHandle(lambda a: 0.0 if a == 0.0 else -5e-4)
    solvers._line_search(h, object())
    assert h.accepted == 1 and len(h.calls) == 1 + 16
    assert np.isclose(h.calls[-1], 0.8 ** 15)
    # immediate success
    h = _FakeHandle(lambda a: 0.0 if a == 0.0 else -2e-3)
    solvers._line_search(h, object())
    assert h.calls == [0.0, 1.0] and h.accepted == 1
