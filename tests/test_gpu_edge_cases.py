"""GPU edge cases through the C-ABI: ragged / empty rows, tiny and odd sizes, widths that exercise every lane
mapping (p = 1 pad column ... p > 128 multi-chunk), dense orders that are not multiples of the MFMA tile,
degenerate constraint sets, and the error paths (call-order violations, unsupported sizes)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


def _pt(n, p, seed):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    return Y, rng.standard_normal((n, p))


@pytest.mark.parametrize("n,p", [(1, 1), (2, 1), (7, 3), (63, 2), (65, 17), (129, 64), (300, 129), (257, 200), (100, 300)])
def test_sparse_ragged_and_empty_rows(lib, n, p):
    from oracle import manisdp_ref as R
    rng = np.random.default_rng(n * 1000 + p)
    A = sp.random(n, n, density=min(1.0, 6.0 / max(n, 1)), random_state=np.random.RandomState(n), format="csr")
    C = (A + A.T).tocsr()
    if n > 3:                              # force some empty rows/columns and one long row
        C = C.tolil(); C[1, :] = 0; C[:, 1] = 0; C[0, :] = 1.0; C[:, 0] = 1.0; C = C.tocsr()
    C.eliminate_zeros()
    Y, U = _pt(n, p, 1)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f = prob.cost(Y)
    assert abs(h.cost() - f) <= 1e-12 * max(1.0, abs(f))
    assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-12 or np.linalg.norm(prob.grad(Y)) < 1e-13
    assert _relerr(h.hessvec(U), prob.hess(Y, U)) < 1e-12
    st = h.rtr(lib.default_opts(maxiter=3, maxinner=5, tolgradnorm=1e-10))
    assert np.isfinite(st.cost) and np.allclose(np.linalg.norm(h.get_point(), axis=1), 1.0, atol=1e-13)
    h.close()


def test_zero_matrix_and_already_optimal_point(lib):
    n, p = 50, 4
    C = sp.csr_matrix((n, n))
    Y, U = _pt(n, p, 0)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    assert h.cost() == 0.0 and np.all(h.rgrad() == 0.0) and np.all(h.hessvec(U) == 0.0)
    st = h.rtr(lib.default_opts(maxiter=5, maxinner=5, tolgradnorm=1e-8))     # gradnorm 0 < tol: stops at k = 0
    assert st.iters == 0 and st.hessvecs == 0 and st.cost == 0.0
    assert np.array_equal(h.get_point(), Y)
    h.close()


@pytest.mark.parametrize("n,p", [(1, 1), (15, 2), (17, 16), (33, 33), (130, 128)])
def test_dense_odd_orders(lib, n, p):
    from oracle import manisdp_ref as R
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)); C = (G + G.T) / 2
    Y, U = _pt(n, p, 2)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f = prob.cost(Y)
    assert abs(h.cost() - f) <= 1e-12 * max(1.0, abs(f))
    assert _relerr(h.hessvec(U), prob.hess(Y, U)) < 1e-12
    h.close()


def test_affine_degenerate_constraints(lib):
    """One constraint only (the trace), an empty constraint column, and a constraint touching a single entry."""
    from oracle import manisdp_ref as R
    n, p = 12, 3
    rows = [i * n + i for i in range(n)] + [5 * n + 5]
    cols = [0] * n + [2]                       # column 1 is empty
    At = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n * n, 3)).tocsc()
    b = np.array([1.0, 0.0, 0.1])
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n)); c = ((G + G.T) / 2).ravel(order="F")
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
    U = rng.standard_normal((n, p))
    y = np.array([0.3, -0.2, 0.1]); sigma = 2.0
    for kind, Prob in ((lib.KIND_UNITTRACE, R._UnitTraceProblem), (lib.KIND_UNITDIAG, R._UnitDiagProblem)):
        Yk = Y if kind == lib.KIND_UNITTRACE else Y / np.linalg.norm(Y, axis=1, keepdims=True)
        prob = Prob(At, b, c, n, p); prob.y, prob.sigma = y, sigma
        f = prob.cost(Yk); Gr = prob.grad(Yk); H = prob.hess(Yk, U)
        h = lib.Handle.affine(kind, At, b, c, n)
        h.set_multipliers(y, sigma)
        h.set_point(Yk)
        assert abs(h.cost() - f) <= 1e-11 * max(1.0, abs(f))
        assert _relerr(h.rgrad(), Gr) < 1e-11 and _relerr(h.hessvec(U), H) < 1e-11
        h.close()


def test_width_changes_between_calls_and_reallocation(lib):
    """The AL loop changes p every iteration; capacity grows on demand and state never leaks between widths."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    C = problems.toroidal_grid_maxcut(12, 11, seed=9)
    n = C.shape[0]
    h = lib.Handle.onlyunitdiag(C, pcap=4)
    for p in (2, 40, 3, 97, 8):
        Y, U = _pt(n, p, p)
        h.set_point(Y)
        prob = R._OnlyUnitDiagProblem(C, n, p); prob.cost(Y)
        assert _relerr(h.hessvec(U), prob.hess(Y, U)) < 1e-12
        st = h.rtr(lib.default_opts(maxiter=2, maxinner=4, tolgradnorm=1e-12))
        assert st.hessvecs > 0
    h.close()


def test_error_paths(lib):
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(6, 6, seed=1)
    h = lib.Handle.onlyunitdiag(C)
    with pytest.raises(lib.MsdpError):            # no resident point yet
        h.rtr(lib.default_opts(maxiter=1, maxinner=1))
    with pytest.raises(lib.MsdpError):
        h.get_point()
    Y, _ = _pt(C.shape[0], 3, 0)
    h.set_point(Y)
    with pytest.raises(lib.MsdpError):            # rho_prime must be < 1/4 (trustregions.m:373-374)
        h.rtr(lib.default_opts(maxiter=1, maxinner=1, rho_prime=0.3))
    with pytest.raises(lib.MsdpError):            # multipliers on a handle without affine constraints
        h.set_multipliers(np.zeros(3), 1.0)
    with pytest.raises(lib.MsdpError):            # p > 1024 unsupported
        h.set_point(np.ones((C.shape[0], 1100)))
    h.close()


@pytest.mark.parametrize("shape,p", [((1, 9), 3), ((5, 10), 5), ((7, 9), 2)])
def test_persistent_tcg_tiny_problems(shape, p):
    """Fewer rows than workgroups / lanes: most workgroups of the persistent tCG kernel own no row at all and
    still have to take part in every grid reduction."""
    from manisdp_matlab_amd import _lib, problems
    from oracle import manisdp_ref as R, manopt_rtr
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=5)
    n = C.shape[0]
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    assert h.tcg_path() == 1
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    for maxinner in (1, 4, 50):
        h.set_point(Y)
        st = h.rtr(_lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        assert st.hessvecs == info.hessvecs
        assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-11 * max(1.0, abs(f_ref))
    h.close()


def test_point_snapshot_restore(lib):
    """msdp_point_snapshot / _restore: restarting from the device-side copy gives exactly the solve that a fresh
    upload of the same point gives."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(20, 30, seed=9)
    n, p = C.shape[0], 6
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = lib.Handle.onlyunitdiag(C, pcap=p)
    opts = lib.default_opts(maxiter=8, maxinner=30, tolgradnorm=1e-9)
    h.set_point(Y)
    h.point_snapshot()
    a = h.rtr(opts)
    Ya = h.get_point()
    h.point_restore()
    assert np.array_equal(h.get_point(), Y)
    b = h.rtr(opts)
    assert (a.cost, a.gradnorm, a.hessvecs, a.iters) == (b.cost, b.gradnorm, b.hessvecs, b.iters)
    assert np.array_equal(h.get_point(), Ya)
    h.set_point(np.hstack([Y, np.zeros((n, 2))]))
    with pytest.raises(lib.MsdpError):
        h.point_restore()                         # snapshot of another width
    h.close()


# ------------------------------------------------------------------ persistent path: recovery and sharing the GPU
def _grid_problem(seed):
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(40, 50, seed=seed)
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((C.shape[0], 16)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    return C, Y


def test_persistent_timeout_falls_back_to_the_chunked_path(lib, capfd):
    """A persistent launch that cannot synchronise its workgroups (here: provoked through the test hook) must not
    surface as an error: msdp_rtr restores the start point, repeats the call on the chunked path inside the same
    process and keeps the handle there."""
    C, Y = _grid_problem(3)
    opts = lib.default_opts(maxiter=25, maxinner=60, tolgradnorm=1e-9)
    h = lib.Handle.onlyunitdiag(C, pcap=16)
    h.set_point(Y)
    assert h.tcg_path() == 1
    ref = h.rtr(opts)                                    # persistent path, undisturbed
    Yref = h.get_point()
    h.set_point(Y)
    h.set_option("debug_fail_persist", 1)
    st = h.rtr(opts)                                     # time-out -> restore -> chunked
    assert "continues on the chunked path" in capfd.readouterr().err
    assert h.tcg_path() == 0                             # the handle stays on the chunked path
    assert abs(st.cost - ref.cost) <= 1e-9 * abs(ref.cost) and abs(st.gradnorm - ref.gradnorm) <= 1e-6 * max(1.0, ref.gradnorm)
    assert _relerr(np.abs(h.get_point() @ h.get_point().T), np.abs(Yref @ Yref.T)) < 1e-6
    # and keeps working from other points without further messages
    h.set_point(Y[:, ::-1].copy())
    st2 = h.rtr(opts)
    assert capfd.readouterr().err == ""
    assert abs(st2.cost - ref.cost) <= 1e-6 * abs(ref.cost)
    h.close()


def test_two_handles_on_two_streams_share_the_gpu(lib):
    """Two handles solving concurrently from two host threads (each handle has its own stream): whichever way the
    hardware interleaves their persistent launches -- both resident, or one starved until its bounded spin gives up and
    the call is repeated on the chunked path -- both calls return the right answer and neither reports an error."""
    import threading
    probs = [_grid_problem(5), _grid_problem(6)]
    opts = lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-9)
    solo = []
    for C, Y in probs:
        h = lib.Handle.onlyunitdiag(C, pcap=16)
        h.set_point(Y)
        solo.append(h.rtr(opts).cost)
        h.close()
    handles = [lib.Handle.onlyunitdiag(C, pcap=16) for C, _ in probs]
    for h, (_, Y) in zip(handles, probs):
        h.set_point(Y)
    out = [None, None]

    def work(i):
        try:
            costs = []
            for _ in range(3):
                handles[i].set_point(probs[i][1])
                costs.append(handles[i].rtr(opts).cost)
            out[i] = costs
        except Exception as e:                            # noqa: BLE001 -- reported below
            out[i] = e
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    for i in range(2):
        assert isinstance(out[i], list), out[i]
        for cst in out[i]:
            assert abs(cst - solo[i]) <= 1e-6 * abs(solo[i])
    for h in handles:
        h.close()


def test_set_option_rejects_unknown_names(lib):
    C, Y = _grid_problem(1)
    h = lib.Handle.onlyunitdiag(C)
    with pytest.raises(lib.MsdpError):
        h.set_option("no_such_option", 1)
    h.set_option("persist", 0)
    h.set_point(Y)
    assert h.tcg_path() == 0
    h.set_option("persist", 1)
    assert h.tcg_path() == 1
    h.close()


def test_factor_gram_rotate_append_on_device():
    """msdp_factor_gram / _rotate / _append against NumPy (the rank decision and the re-shaping of the factor
    between two trustregions() calls happen on the device), odd and even widths, and the capacity error."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(20, 31, seed=2)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    for p, r, k in ((7, 4, 3), (12, 12, 1), (5, 1, 8)):
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = _lib.Handle.onlyunitdiag(C, pcap=32)
        h.set_point(Y)
        G = h.factor_gram()
        assert np.linalg.norm(G - Y.T @ Y) <= 1e-12 * np.linalg.norm(Y.T @ Y)
        Q = np.linalg.qr(rng.standard_normal((p, r)))[0]
        h.factor_rotate(Q)
        Yr = h.get_point()
        assert Yr.shape == (n, r) and np.linalg.norm(Yr - Y @ Q) <= 1e-13 * np.linalg.norm(Y @ Q)
        V = rng.standard_normal((n, k))
        h.factor_append(V, 0.3, normalize=True)
        Ya = h.get_point()
        ref = np.hstack([Y @ Q, 0.3 * V]); ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        assert Ya.shape == (n, r + k) and np.linalg.norm(Ya - ref) <= 1e-13 * np.linalg.norm(ref)
        # the re-shaped point is a valid resident point: cost / gradient follow
        h2 = _lib.Handle.onlyunitdiag(C, pcap=32)
        h2.set_point(ref)
        assert abs(h.cost() - h2.cost()) <= 1e-12 * abs(h2.cost())
        assert np.linalg.norm(h.rgrad() - h2.rgrad()) <= 1e-12 * np.linalg.norm(h2.rgrad())
        with pytest.raises(_lib.MsdpError, match="allocated capacity"):
            h.factor_append(rng.standard_normal((n, 40)), 0.1)
        h.close(); h2.close()


def test_device_factor_path_follows_host_path():
    """ManiSDP_onlyunitdiag with the factor kept on the device between the trustregions() calls against the host form of the
    same steps (G1, device escape): same number of outer iterations, same optimum."""
    from manisdp_matlab_amd import problems, solvers
    from conftest import golden_path
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    rng = np.random.default_rng(3)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    out = []
    for dev in (True, False):
        Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "eig": "device", "device_factor": dev, "tol": 1e-8}, verbose=False)
        assert data["status"] == 0 and data["dinf"] < 1e-8
        assert np.allclose(np.linalg.norm(Y, axis=1), 1.0, atol=1e-12)
        out.append((obj, data["iters"], Y.shape[1]))
    assert abs(out[0][0] - out[1][0]) <= 1e-7 * abs(out[1][0])
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


@pytest.mark.parametrize("rows,cols,p", [(100, 200, 40), (150, 160, 40), (150, 160, 32), (130, 200, 32), (100, 200, 16),
                                         (200, 200, 32), (250, 250, 24), (200, 200, 16), (250, 250, 12)])
def test_persistent_instances_agree_with_chunked_path(rows, cols, p):
    """Every row-slot instance of the persistent tCG kernel (two / four slots at p <= 16, three / four / eight at
    p = 17..32, five / eight at p = 33..64, chosen by the rows a workgroup owns; n up to 62 500 here) against the chunked
    three-kernel path on the same start point: same Hess-vec count and
    stop decisions, cost and gradient norm to rounding."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(rows, cols, seed=3)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    res = []
    for persist in (1, 0):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", persist)
        h.set_point(Y)
        assert h.tcg_path() == persist
        st = h.rtr(_lib.default_opts(maxiter=6, maxinner=40, tolgradnorm=1e-9))
        res.append((st.hessvecs, st.accepted, st.rejected, st.cost, st.gradnorm, h.get_point()))
        h.close()
    a, b = res
    assert a[:3] == b[:3]
    assert abs(a[3] - b[3]) <= 1e-11 * abs(b[3]) and abs(a[4] - b[4]) <= 1e-7 * max(b[4], 1e-12)
    assert np.linalg.norm(a[5] - b[5]) <= 1e-8 * np.linalg.norm(b[5])


@pytest.mark.parametrize("p", [600, 1000])
def test_wide_factors_up_to_1024(p):
    """Factor widths beyond 512 (eight column chunks per lane; a quartic-on-sphere relaxation with d = 100 asks for p = 518):
    sparse C, dense C and the unit-diagonal affine operators against the oracle; p > 1024 is refused."""
    import scipy.sparse as sp
    from manisdp_matlab_amd import _lib, problems
    from oracle import manisdp_ref as R
    from conftest import golden_path
    _lib.load()
    rng = np.random.default_rng(p)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)    # noqa: E731
    # sparse and dense onlyunitdiag
    C = problems.toroidal_grid_maxcut(30, 40, seed=1)
    n = C.shape[0]
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f, G, H = prob.cost(Y), prob.grad(Y), prob.hess(Y, U)
    for Cin in (C, C.toarray()):
        h = _lib.Handle.onlyunitdiag(Cin, pcap=p)
        h.set_point(Y)
        assert abs(h.cost() - f) <= 1e-11 * abs(f)
        assert rel(h.rgrad(), G) < 1e-11 and rel(h.hessvec(U), H) < 1e-11
        st = h.rtr(_lib.default_opts(maxiter=2, maxinner=5, tolgradnorm=1e-9))
        assert st.hessvecs > 0 and np.isfinite(st.cost)
        h.close()
    # unit-diagonal affine kind (BQP d = 10)
    Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(10, Q, e)
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    n = K["s"]
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p)); U -= Y * np.sum(Y * U, axis=1, keepdims=True)
    y = 0.1 * rng.standard_normal(b.size)
    pa = R._UnitDiagProblem(At, b, c, n, p)
    pa.y, pa.sigma = y, 0.7
    fa, Ga, Ha = pa.cost(Y), pa.grad(Y), pa.hess(Y, U)
    h = _lib.Handle.affine(_lib.KIND_UNITDIAG, sp.csc_matrix(At), b, c, n, pcap=p)
    h.set_multipliers(y, 0.7)
    h.set_point(Y)
    assert abs(h.cost() - fa) <= 1e-10 * abs(fa)
    assert rel(h.rgrad(), Ga) < 1e-10 and rel(h.hessvec(U), Ha) < 1e-10
    with pytest.raises(_lib.MsdpError, match="maximum of 1024"):
        h.set_point(np.ones((n, 1030)) / np.sqrt(1030.0))
    h.close()
    # unit-trace affine kind (theta1): the fused sphere epilogue with eight column chunks per lane (round 5: holds the sparse part only)
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float).ravel()
    n = K["s"]
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
    pt = R._UnitTraceProblem(At, b, c, n, p)
    U = pt.M.proj(Y, rng.standard_normal((n, p)))
    y = 0.1 * rng.standard_normal(b.size)
    pt.y, pt.sigma = y, 2.5
    ft, Gt, Ht = pt.cost(Y), pt.grad(Y), pt.hess(Y, U)
    h = _lib.Handle.affine(_lib.KIND_UNITTRACE, At, b, c, n, pcap=p)
    h.set_multipliers(y, 2.5)
    h.set_point(Y)
    assert abs(h.cost() - ft) <= 1e-10 * max(1.0, abs(ft))
    assert rel(h.rgrad(), Gt) < 1e-10 and rel(h.hessvec(U), Ht) < 1e-10
    h.close()


def test_handle_churn_keeps_results_and_pool_bounded(lib):
    """VERDICT round 3, item 8.  The words that workgroups exchange inside a launch (grid-sync slots, direction rows) live in
    uncached device memory; round 3 found that uncached blocks allocated and hipFree'd per handle corrupt LATER handles of the
    process.  The fence: uncached memory comes from per-process arenas with a coalescing sub-allocator (msdp_api.hip) that
    never hand pages back to the driver while any block of them is live.  Here 200 handles of mixed kinds and sizes are created
    and destroyed with overlapping lifetimes; every one must reproduce the oracle's cost, eG / z and gradient and a short
    trustregions() call through the persistent kernels (the users of the uncached blocks), the pool must stay below
    2 x (largest number of bytes live together) + two arenas, and nothing may be live when the last handle is gone."""
    import json
    from conftest import golden_path
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    rng = np.random.default_rng(11)
    grids = [(6, 7), (24, 32), (50, 61), (100, 100), (141, 142), (100, 300)]
    sparse = [problems.toroidal_grid_maxcut(r, c, seed=r) for r, c in grids]
    dense = [problems.dense_unitdiag_cost(n, seed=n) for n in (33, 200, 600)]
    At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float).ravel()
    Atg, bg, cg, Kg = problems.from_sdpa(golden_path("gpp100.dat-s.gz"))
    cg = np.asarray(cg.todense()).ravel(); bg = np.asarray(bg, float).ravel()
    lib.release_cache()
    base_pool = lib.pool_stats()[0]
    alive, peak_live, peak_pool = [], 0, 0

    def check_sparse_or_dense(C, p, seed):
        n = C.shape[0]
        Y, U = _pt(n, p, seed)
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_point(Y)
        CY = C @ Y
        z = np.sum(CY * Y, axis=1)
        assert abs(h.cost() - 0.5 * z.sum()) <= 1e-12 * max(1.0, abs(0.5 * z.sum()))
        assert _relerr(h.get_z(), z) < 1e-12
        assert _relerr(h.rgrad(), CY - Y * z[:, None]) < 1e-12
        st = h.rtr(lib.default_opts(maxiter=3, maxinner=12, tolgradnorm=1e-9))
        assert np.isfinite(st.cost) and st.cost <= 0.5 * z.sum() + 1e-9 * abs(z.sum())
        assert np.abs(np.linalg.norm(h.get_point(), axis=1) - 1.0).max() < 1e-12
        return h

    def check_affine(kind, At_, b_, c_, n, p, seed):
        r = np.random.default_rng(seed)
        Y = r.standard_normal((n, p))
        Y /= np.linalg.norm(Y, axis=1, keepdims=True) if kind == lib.KIND_UNITDIAG else np.linalg.norm(Y)
        prob = (R._UnitDiagProblem if kind == lib.KIND_UNITDIAG else R._UnitTraceProblem)(At_, b_, c_, n, p)
        y = r.standard_normal(b_.size) * 0.1
        prob.y, prob.sigma = y, 1.3
        h = lib.Handle.affine(kind, At_, b_, c_, n)
        h.set_multipliers(y, 1.3)
        h.set_point(Y)
        f_ref = prob.cost(Y)
        assert abs(h.cost() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
        assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-11
        return h

    for it in range(200):
        kind = it % 10
        if kind < 6:
            h = check_sparse_or_dense(sparse[int(rng.integers(len(sparse)))], int(rng.integers(2, 41)), it)
        elif kind < 8:
            h = check_sparse_or_dense(dense[int(rng.integers(len(dense)))], int(rng.integers(2, 33)), it)
        elif kind == 8:
            h = check_affine(lib.KIND_UNITTRACE, At, b, c, K["s"], int(rng.integers(1, 12)), it)
        else:
            h = check_affine(lib.KIND_UNITDIAG, Atg, bg, cg, Kg["s"], int(rng.integers(2, 12)), it)
        alive.append(h)
        while len(alive) > int(rng.integers(1, 6)):            # overlapping lifetimes, closed in random order
            alive.pop(int(rng.integers(len(alive)))).close()
        pool, live, arenas = lib.pool_stats()
        peak_live, peak_pool = max(peak_live, live), max(peak_pool, pool)
        assert live <= pool
    for h in alive:
        h.close()
    pool, live, arenas = lib.pool_stats()
    assert live == 0
    assert peak_pool - base_pool <= 2 * peak_live + 2 * (32 << 20), (peak_pool, peak_live)
    # round 5: the arenas are fine-grained memory (safe to hand back: tools/uc_pool_stress.py mode 6); with MSDP_UC_MEM=uncached
    # they stay with the process for good (pages that were uncached are not safe in anybody else's hands)
    lib.release_cache()
    assert lib.pool_stats()[0] == (pool if os.environ.get("MSDP_UC_MEM") == "uncached" else 0)
