"""CPU tests (no GPU): pin the oracle against every known answer the reference ships for this
path (data/sdplib/README:39-51,71-88,98-105 -> tests/golden/known_answers.json) and against the
SURVEY.md probe value for G1.  These are what make the oracle trustworthy as the GPU checker."""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bqp_bruteforce_min, golden_path, within_print
from manisdp_matlab_amd import problems
from oracle import manisdp_ref as R

KNOWN = json.load(open(golden_path("known_answers.json")))
PRINTED = json.load(open(golden_path("known_answers_printed.json")))       # the same values as data/sdplib/README prints them

# Option sets under which the reference's algorithm reaches KKT 1e-8 on the SDPLIB families whose defaults are tuned for
# other problems (found on the oracle, tools/archive/theta_ref_opts.py): a larger trust-region budget per outer iteration.
THETA_OPTS = dict(tol=1e-8, TR_maxiter=30, TR_maxinner=200)
GPP_OPTS = dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6)


def _mcp(name):
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    n = K["s"]
    return sp.csr_matrix(c.toarray().reshape(n, n, order="F"))


@pytest.mark.parametrize("name", ["mcp100", "mcp124-1", "mcp124-2", "mcp124-3", "mcp124-4", "mcp250-1", "mcp250-2", "mcp250-3",
                                  "mcp250-4", "mcp500-1", "mcp500-2", "mcp500-3", "mcp500-4"])
def test_onlyunitdiag_sdplib_mcp(name):
    """max <F0,X>, X_ii = 1  ->  ManiSDP_onlyunitdiag with C = -F0 returns -value: every mcp instance of
    data/sdplib/README:76-88, to the 7 digits printed there."""
    Y, obj, data = R.ManiSDP_onlyunitdiag(_mcp(name), {})
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert within_print(-obj, PRINTED[name])
    assert np.allclose(np.linalg.norm(Y, axis=1), 1.0, atol=1e-12)


def test_onlyunitdiag_maxG32():
    """maxG32 == Gset G32 (n = 2000) with C = -L/4: data/sdplib/README:72, 7 digits."""
    Y, obj, data = R.ManiSDP_onlyunitdiag(problems.maxcut_cost_matrix(golden_path("G32.txt.gz")), {})
    assert data["status"] == 0 and data["dinf"] < 1e-8
    assert within_print(-obj, PRINTED["maxG32"])


def test_onlyunitdiag_gset():
    """maxG11 == Gset G11 with C = -L/4 (example_maxcut.m:10-11); G1 against the survey probe."""
    Y, obj, data = R.ManiSDP_onlyunitdiag(problems.maxcut_cost_matrix(golden_path("G11.txt.gz")), {})
    assert data["dinf"] < 1e-8
    assert within_print(-obj, PRINTED["maxG11"])
    Y, obj, data = R.ManiSDP_onlyunitdiag(problems.maxcut_cost_matrix(golden_path("G1.txt.gz")), {})
    assert data["dinf"] < 1e-8
    assert abs(obj - (-12083.19765455)) < 1e-6 * 12083.2


def test_quirk_q1_variants_agree_at_optimum():
    """Reference behaviour (stale eG after a rejected step) and per-point state reach the same optimum."""
    C = problems.maxcut_cost_matrix(golden_path("G11.txt.gz"))
    rng = np.random.default_rng(1)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    _, o1, d1 = R.ManiSDP_onlyunitdiag(C, {"Y0": Y0}, q1="reference")
    _, o2, d2 = R.ManiSDP_onlyunitdiag(C, {"Y0": Y0}, q1="correct")
    assert d1["dinf"] < 1e-8 and d2["dinf"] < 1e-8
    assert abs(o1 - o2) < 1e-7 * abs(o1)


@pytest.mark.parametrize("name", ["gpp100", "gpp124-1", "gpp124-2", "gpp124-3", "gpp124-4"])
def test_unitdiag_gpp(name):
    """gpp* through fromsdpa: constraint 1 is <J,X> = 0, the rest X_ii = 1 (data/sdplib/README:39-43, 5-6 digits printed:
    the value must agree to exactly those digits).  Option set of SURVEY.md section 4 (the defaults are tuned for BQP); the
    family crawls at KKT residues of 1e-7 (no strictly feasible point: <J,X> = 0 with X psd), so the run ends on the
    reference's iteration limit -- the value is converged to the printed digits long before."""
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    Y, obj, data = R.ManiSDP_unitdiag(At, b, c, K, dict(GPP_OPTS))
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-6
    assert within_print(-obj, PRINTED[name])


def test_unitdiag_bqp_kkt_self_certification():
    """BQP instances have no stored optimum: KKT residues below tol certify the value."""
    Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(10, Q, e)
    c = np.asarray(c.todense()).ravel()
    Y, obj, data = R.ManiSDP_unitdiag(At, b, c / np.abs(c).max(), K, {})
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    X = Y @ Y.T
    assert np.allclose(np.diag(X), 1.0, atol=1e-12)
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) / (1 + np.linalg.norm(b)) < 1e-8


@pytest.mark.parametrize("name", ["theta1", "theta2", "theta3", "theta4", "theta5"])
def test_unittrace_theta(name):
    """theta1..theta4 (data/sdplib/README:98-101, 7 digits): constraint 1 is tr X = 1, F0 = J.  With the default trust-region
    budget (3 x 40) the reference's algorithm leaves through "Slow progress" at KKT residues of 1e-5..1e-4 with the value right
    to 2e-6 only; with 30 x 200 it converges to tol = 1e-8 in ~20 outer iterations and the value matches all printed digits."""
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    Y, obj, data = R.ManiSDP_unittrace(At, b, c, K, dict(THETA_OPTS))
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert within_print(-obj, PRINTED[name])
    assert abs(-obj - KNOWN[name]) < 1e-7 * KNOWN[name]
    assert abs(np.linalg.norm(Y) - 1.0) < 1e-12


def test_unittrace_theta1_reference_example_options():
    """theta1 with the options of example/example_theta.m:50-53 (tol 1e-6): converges for this start."""
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    Y, obj, data = R.ManiSDP_unittrace(At, b, c, K, dict(tol=1e-6, sigma0=1e5, sigma_max=1e8))
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-5
    assert abs(-obj - KNOWN["theta1"]) < 1e-5 * KNOWN["theta1"]


@pytest.mark.parametrize("name", ["mcp100", "mcp124-1"])
def test_generic_sdplib_mcp(name):
    """The generic entry point (src/primal/ManiSDP.m, Euclidean manifold) on the same SDPLIB instances, treated
    as plain A(X) = b problems: same README values."""
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    Y, obj, data = R.ManiSDP(At, b, c, K, {})
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert within_print(-obj, PRINTED[name])


def test_generic_quartic_on_sphere_kkt_self_certification():
    """example/example_qsphere.m:18-27 with the reference's coefficient file data/qs_c_10_1.txt: the reference stores
    no optimum for these instances, so the result is pinned by the KKT residues (gap, pinf, dinf < 1e-8 certify the
    objective to ~1e-8 by weak duality) and by the constraints A(X) = b of the moment matrix X = Y Y'."""
    import numpy as np
    coe = np.loadtxt(golden_path("qs_c_10_1.txt.gz"), delimiter=",").ravel()
    At, b, c, K = problems.qsmom(10, coe)
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    Y, obj, data = R.ManiSDP(At, b, c, K, {})
    assert data["status"] == 0
    assert max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    X = Y @ Y.T
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) / (1 + np.linalg.norm(b)) < 1e-8
    assert abs(obj - c @ X.ravel(order="F")) < 1e-9 * max(1.0, abs(obj))


def test_multiblock_direct_sum_of_two_maxcut_problems():
    """ManiSDP_multiblock.m restated (oracle): mcp100 (+) mcp124-1 as one two-block SDP with both unit diagonals in the
    product manifold has the sum of the two SDPLIB optima (data/sdplib/README:76-77)."""
    import scipy.sparse as sp
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    known = json.load(open(golden_path("known_answers.json")))
    cs, ns = [], []
    for name in ("mcp100", "mcp124-1"):
        At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
        cs.append(np.asarray(c.todense()).ravel()); ns.append(K["s"])
    c = np.concatenate(cs)
    At = sp.csc_matrix((np.ones(1), ([0], [0])), shape=(c.size, 1))
    Y, obj, d = R.ManiSDP_multiblock(At, np.ones(1), c, {"s": ns, "nob": 2}, {})
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    want = -(known["mcp100"] + known["mcp124-1"])
    assert abs(obj - want) <= 1e-6 * abs(want)


def test_dual_oracle_agrees_with_primal_oracle():
    """ManiDSDP_unitdiag on the SOS relaxation of a BQP (bqpsos.m data as example_bqp_dual.m builds it) and
    ManiSDP_unitdiag on the moment relaxation of the same BQP (bqpmom.m) are a primal/dual pair: same optimum.  Pins the
    dual restatement to the primal one, which is pinned to the SDPLIB values above."""
    import numpy as np
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    for d, seed in ((6, 0), (9, 4)):
        rng = np.random.default_rng(seed)
        Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
        e = rng.standard_normal(d)
        At, b, c, K = problems.bqpmom(d, Q, e)
        _, fp, dp = R.ManiSDP_unitdiag(At, b, c, K, {"tol": 1e-8}, rng=np.random.default_rng(0))
        A, bs, cs, Ks, dAAt, maxb = problems.bqpsos_dual_problem(Q, e, d)
        assert Ks["s"] == K["s"] and np.allclose(dAAt, np.asarray(A[:, 1:].multiply(A[:, 1:]).sum(axis=1)).ravel())
        for ls in (0, 1):
            _, fd, dd = R.ManiDSDP_unitdiag(A, bs, cs, Ks, {"tol": 1e-8, "dAAt": dAAt, "line_search": ls}, rng=np.random.default_rng(0))
            assert dd["status"] == 0 and max(dd["gap"], dd["pinf"], dd["dinf"]) < 1e-8
            assert abs(fd * maxb - fp) <= 1e-6 * max(1.0, abs(fp))


@pytest.mark.parametrize("d", [10])
def test_bqp_relaxation_is_bounded_by_the_bruteforce_minimum(d):
    """A pin of bqpmom + ManiSDP_unitdiag that is independent of any restatement: the second-order moment relaxation
    (src/basicfunction/bqpmom.m:1-5: Min x'Qx + x'e, x_i^2 = 1) is a LOWER bound of the minimum over the 2^d sign vectors,
    and when the returned X has rank one its first column IS a sign vector attaining the value, so the two are equal.
    Instance: data/bqp_Q_10_1.txt / bqp_e_10_1.txt, scaled as example/example_bqp.m:39-41 does."""
    Q = np.loadtxt(golden_path(f"bqp_Q_{d}_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path(f"bqp_e_{d}_1.txt.gz"), delimiter=",")
    fmin, xmin = bqp_bruteforce_min(Q, e)
    At, b, c, K = problems.bqpmom(d, Q, e)
    c = np.asarray(c.todense()).ravel()
    maxc = np.abs(c).max()
    Y, obj, data = R.ManiSDP_unitdiag(At, b, c / maxc, K, {"tol": 1e-8})
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    val = obj * maxc
    assert val <= fmin + 1e-6 * max(1.0, abs(fmin))
    sv = np.linalg.svd(Y, compute_uv=False)
    if sv[1] < 1e-6 * sv[0]:                                   # rank one: the relaxation is tight
        assert abs(val - fmin) <= 1e-6 * max(1.0, abs(fmin))
        X = Y @ Y.T
        x = np.sign(X[1:d + 1, 0])
        assert abs(x @ Q @ x + e @ x - fmin) <= 1e-9 * max(1.0, abs(fmin))


def test_unitdiag_thetaG11():
    """data/sdplib/README:104 (thetaG11: m = 2401, n = 801, 4.000000e+02).  Listed with the theta family, but the instance is
    a unit-DIAGONAL problem -- constraints 1..801 are X_ii = 1, the other 1600 tie one 3 x 3 all-ones pattern per edge of
    Gset G11 to 1 -- so it pins ManiSDP_unitdiag (default options) at n = 801, to the 7 printed digits."""
    At, b, c, K = problems.from_sdpa(golden_path("thetaG11.dat-s.gz"))
    At = At.tocsc()
    n = K["s"]
    assert (n, At.shape[1]) == (801, 2401)
    for k in range(n):                                        # the structure the choice of entry point rests on
        col = At[:, k].tocoo()
        assert col.nnz == 1 and col.row[0] == k * n + k and col.data[0] == 1.0
    c = np.asarray(c.todense()).ravel()
    Y, obj, data = R.ManiSDP_unitdiag(At, np.asarray(b).ravel(), c, K, {"tol": 1e-8})
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert within_print(-obj, PRINTED["thetaG11"])


@pytest.mark.parametrize("k,d,theta", [(3, [1, 2, 3], 1.0), (4, [1], 8.0), (4, [4], 8.0), (5, [1], 16.0)])
def test_unittrace_hamming_graphs_with_known_theta(k, d, theta):
    """example/generate_hamming.m (the generator of SDPLIB's hamming_* problems, restated in problems.generate_hamming) through
    ManiSDP_unittrace on graphs whose theta number is known in closed form: the complete graph (distances 1..k: theta = 1) and
    perfect graphs -- the hypercube (distance 1, bipartite) and the antipodal matching (distance k) -- where theta = the
    independence number 2^(k-1).  A pin of generator + unit-trace solver that needs no stored optimum."""
    At, b, c, K = problems.generate_hamming(k, d)
    n = 1 << k
    assert K["s"] == n and b[0] == 1.0 and np.count_nonzero(b) == 1
    col0 = At[:, 0].tocoo()
    assert col0.nnz == n and np.array_equal(np.sort(col0.row), np.arange(n) * (n + 1))          # constraint 1 = trace
    Y, obj, data = R.ManiSDP_unittrace(At, b, c, K, dict(THETA_OPTS))
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    assert abs(-obj - theta) <= 1e-7 * theta


QUASAR_OPTS = {"tol": 1e-8, "sigma0": 1.0, "sigma_min": 1.0, "sigma_max": 1e4}


def check_rotation_search(solve, N, seed):
    """Shared by the oracle and the GPU test: the QUASAR relaxation of a rotation search with 50 % outliers is tight -- the optimum
    of the SDP equals the truncated-least-squares cost AT the rotation read off its rank-one solution (an independent evaluation:
    3 x 3 rotation, N residuals), the inlier set comes back exactly, and the rotation is the ground truth to the noise level."""
    a, b, Rgt, beta, out = problems.wahba_with_outliers(N, 0.5, seed=seed)
    At, bv, c, K = problems.quasar_problem(a, b, beta ** 2)
    Y, fval, data = solve(At, bv / (N + 1), c, K, dict(QUASAR_OPTS))          # example_rotationsearch.m:37: b / (N + 1)
    assert data["status"] == 0 and max(data["gap"], data["pinf"], data["dinf"]) < 1e-8
    X = Y @ Y.T
    w = np.linalg.eigvalsh(X)
    assert w[-2] <= 1e-8 * w[-1]                                               # rank one
    Rr, theta = problems.quasar_recover(X, N)
    assert np.array_equal(theta > 0, ~out)
    tls = sum(min(np.sum((bi - Rr @ ai) ** 2) / beta ** 2, 1.0) for ai, bi in zip(a, b))
    assert abs(fval * (N + 1) - tls) <= 1e-6 * tls
    angle = np.degrees(np.arccos(np.clip((np.trace(Rr.T @ Rgt) - 1.0) / 2.0, -1.0, 1.0)))
    assert angle < 1.0
    return fval


def test_unittrace_rotation_search_relaxation_is_tight():
    """example/example_rotationsearch.m through the oracle's ManiSDP_unittrace (N = 10 measurements, half of them outliers).  The
    reference's sigma schedule (sigma_min = 1e2, up to 1e7) leaves through "Slow progress" on this data (the example's own generator
    is not in the tree, its scaling unknown); sigma0 = sigma_min = 1, sigma_max = 1e4 converges in ~50 outer iterations."""
    check_rotation_search(R.ManiSDP_unittrace, 10, 1)

