"""The N-rank code paths on ONE GPU: N handles of this process, one host thread each, joined by the in-process stand-in for the
communicator (msdp_comm_init_local: same row partition, same lock-step tCG driver, same order and number of collective calls as
the RCCL run; the collectives themselves are a host barrier plus device copies / a summation kernel).  Every test compares
the ranks' combined result with one unsharded handle.  What stays untested on this box is RCCL itself with more than one
member -- tests/test_gpu_comm.py runs that wherever N GPUs are visible."""
import threading

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_path

pytestmark = pytest.mark.gpu
_group = [1000]


def run_ranks(N, fn):
    """fn(rank, group) on N threads; returns the list of results, re-raises the first exception."""
    _group[0] += 1
    group = _group[0]
    out, err = [None] * N, [None] * N
    from manisdp_matlab_amd import solvers
    solvers.FORCED_HOST_THREADS = 1          # N replicated host loops share this process's BLAS pool: see solvers.py

    def body(r):
        try:
            out[r] = fn(r, group)
        except BaseException as e:      # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(N)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    solvers.FORCED_HOST_THREADS = None
    for e in err:
        if e is not None:
            raise e
    return out


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("N", [2, 3, 8])
@pytest.mark.parametrize("case", ["sparse", "dense", "sparse-halo"])
def test_onlyunitdiag_ranks_match_one_handle(N, case):
    halo = case == "sparse-halo"                   # option halo_exchange: only the referenced rows travel before S*U
    case = case.split("-")[0]
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    rng = np.random.default_rng(11)
    if case == "sparse":
        C = problems.toroidal_grid_maxcut(61, 50, seed=4)           # n = 3050: ragged last shard for N = 3, 8
        n, p = C.shape[0], 12
    else:
        n, p = 1000, 24
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    opts = _lib.default_opts(maxiter=8, maxinner=25, tolgradnorm=1e-9)

    def make(nranks=1, rank=0):
        return _lib.Handle.onlyunitdiag(C, pcap=p) if case == "sparse" else _lib.Handle.dense_synthetic(n, 3, nranks=nranks, rank=rank, pcap=p)

    def one_rank(r, group):
        h = make(N, r)
        h.comm_init_local(N, r, group)
        if halo:
            h.set_option("halo_exchange", 1)
        r0, r1 = h.local_rows()
        h.set_point(Y)
        f = h.cost(); G = h.rgrad(); H = h.hessvec(U)
        st = h.rtr(opts)
        Yall = h.get_point_all()
        z = h.get_z_all()
        h.close()
        return dict(rows=(r0, r1), f=f, G=G[r0:r1], H=H[r0:r1], cost=st.cost, gradnorm=st.gradnorm, hessvecs=st.hessvecs,
                    accepted=st.accepted, rejected=st.rejected, Y=Yall, z=z)

    res = run_ranks(N, one_rank)
    h = make()
    h.set_option("persist", 0)                            # the sharded run uses the chunked kernels
    h.set_point(Y)
    f = h.cost(); G = h.rgrad(); H = h.hessvec(U)
    st = h.rtr(opts)
    Yout = h.get_point(); z = h.get_z()
    h.close()
    assert res[0]["rows"][0] == 0 and res[-1]["rows"][1] == n
    Gs = np.vstack([q["G"] for q in res]); Hs = np.vstack([q["H"] for q in res])
    assert rel(Gs, G) < 1e-12 and rel(Hs, H) < 1e-12
    for q in res:
        assert abs(q["f"] - f) <= 1e-12 * abs(f)
        assert (q["hessvecs"], q["accepted"], q["rejected"]) == (st.hessvecs, st.accepted, st.rejected)
        assert abs(q["cost"] - st.cost) <= 1e-10 * abs(st.cost)
        assert rel(q["Y"], Yout) < 1e-8 and rel(q["z"], z) < 1e-8
        assert np.array_equal(q["Y"], res[0]["Y"]) and np.array_equal(q["z"], res[0]["z"])       # replicated data is identical


def _bqp10():
    from manisdp_matlab_amd import problems
    Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(10, Q, e)
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    At = sp.csc_matrix(At); At.sort_indices()
    return At, b, c, K


@pytest.mark.parametrize("N", [2, 3])
@pytest.mark.parametrize("kind_name", ["unitdiag", "unittrace"])
def test_affine_ranks_match_one_handle(N, kind_name):
    """Row-sharded affine kinds with the replicated operator state: operators, line-search cost, trustregions(), AL
    bookkeeping and the replicated escape on N in-process ranks against one handle."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    if kind_name == "unitdiag":
        At, b, c, K = _bqp10()
        kind = _lib.KIND_UNITDIAG
    else:
        At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
        c = np.asarray(c.todense()).ravel() if sp.issparse(c) else np.asarray(c, float).ravel()
        b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
        At = sp.csc_matrix(At); At.sort_indices()
        kind = _lib.KIND_UNITTRACE
    n, m, p = K["s"], b.size, 6
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p))
    Y = Y / np.linalg.norm(Y, axis=1, keepdims=True) if kind == _lib.KIND_UNITDIAG else Y / np.linalg.norm(Y)
    U = 0.3 * rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(m)
    opts = _lib.default_opts(maxiter=3, maxinner=15, tolgradnorm=1e-8)

    def session(h, all_rows):
        h.set_multipliers(y, 0.7)
        h.set_point(Y)
        f = h.cost(); G = h.rgrad(); H = h.hessvec(h.proj(U))
        co = h.linesearch_cost(U, 0.5)
        st = h.rtr(opts)
        Yr = h.get_point_all() if all_rows else h.get_point()
        obj, Ax = h.al_primal(m)
        z = h.al_dual(y)
        lam, V, lmax, _ = h.escape_eigs_dual(3, tol=1e-10, maxit=4000)
        return dict(f=f, G=G, H=H, co=co, cost=st.cost, hessvecs=st.hessvecs, Y=Yr, obj=obj, Ax=Ax, z=np.atleast_1d(z), lam=lam, lmax=lmax)

    def one_rank(r, group):
        h = _lib.Handle.affine(kind, At, b, c, n)
        h.comm_init_local(N, r, group)
        r0, r1 = h.local_rows()
        q = session(h, True)
        h.close()
        q["G"], q["H"], q["rows"] = q["G"][r0:r1], q["H"][r0:r1], (r0, r1)
        return q

    res = run_ranks(N, one_rank)
    h = _lib.Handle.affine(kind, At, b, c, n)
    ref = session(h, False)
    h.close()
    assert rel(np.vstack([q["G"] for q in res]), ref["G"]) < 1e-10
    assert rel(np.vstack([q["H"] for q in res]), ref["H"]) < 1e-10
    for q in res:
        assert abs(q["f"] - ref["f"]) <= 1e-11 * abs(ref["f"]) and abs(q["co"] - ref["co"]) <= 1e-11 * abs(ref["co"])
        assert q["hessvecs"] == ref["hessvecs"] and abs(q["cost"] - ref["cost"]) <= 1e-9 * abs(ref["cost"])
        assert rel(q["Y"], ref["Y"]) < 1e-7 and rel(q["Ax"], ref["Ax"]) < 1e-7 and rel(q["z"], ref["z"]) < 1e-7
        assert abs(q["obj"] - ref["obj"]) <= 1e-8 * max(1.0, abs(ref["obj"]))
        ok = np.isfinite(ref["lam"])
        assert np.array_equal(np.isfinite(q["lam"]), ok) and rel(q["lam"][ok], ref["lam"][ok]) < 1e-5
        assert abs(q["lmax"] - ref["lmax"]) <= 1e-6 * abs(ref["lmax"])
        for key in ("f", "co", "cost", "obj", "lmax"):                                   # replicated scalars: identical bits
            assert q[key] == res[0][key], key
        assert np.array_equal(q["Y"], res[0]["Y"]) and np.array_equal(q["Ax"], res[0]["Ax"])


@pytest.mark.parametrize("N", [2, 4])
def test_whole_solves_on_in_process_ranks(N):
    """The replicated host loops with options['comm'] = ('local', N, rank, group): ManiSDP_onlyunitdiag (sharded RTR, escape on
    the replicated copy of C, device-resident factor) and ManiSDP_unitdiag (BQP d = 10) against the one-handle solves."""
    from manisdp_matlab_amd import problems, solvers
    C = problems.toroidal_grid_maxcut(30, 40, seed=6)
    rng = np.random.default_rng(9)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    At, b, c, K = _bqp10()
    Z0 = rng.standard_normal((K["s"], 2)); Z0 /= np.linalg.norm(Z0, axis=1, keepdims=True)

    def one_rank(r, group):
        Y, obj, d = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "tol": 1e-8, "comm": ("local", N, r, group)}, verbose=False)
        Ya, obja, da = solvers.ManiSDP_unitdiag(At, b, c, K, {"Y0": Z0, "tol": 1e-8, "comm": ("local", N, r, group + 500)}, verbose=False)
        return (obj, d["status"], d["dinf"], Y, obja, da["status"], max(da["gap"], da["pinf"], da["dinf"]))

    res = run_ranks(N, one_rank)
    _, obj1, d1 = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "tol": 1e-8, "eig": "device"}, verbose=False)
    _, obja1, da1 = solvers.ManiSDP_unitdiag(At, b, c, K, {"Y0": Z0, "tol": 1e-8}, verbose=False)
    assert d1["status"] == 0 and da1["status"] == 0
    for q in res:
        assert q[1] == 0 and q[2] < 1e-8 and abs(q[0] - obj1) <= 1e-7 * abs(obj1)
        assert np.allclose(np.linalg.norm(q[3], axis=1), 1.0, atol=1e-12)
        assert q[5] == 0 and q[6] < 1e-8 and abs(q[4] - obja1) <= 1e-6 * max(1.0, abs(obja1))
        assert q[0] == res[0][0] and q[4] == res[0][4]                                     # every rank reports the same numbers


def test_ragged_partition_with_an_empty_rank():
    """n = 20 over 8 ranks: blocks of 3 rows, the seventh rank owns 2, the eighth none -- and the solve is the unsharded one."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(5, 4, seed=1)
    n, p, N = C.shape[0], 3, 8
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=5, maxinner=10, tolgradnorm=1e-9)

    def one_rank(r, group):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.comm_init_local(N, r, group)
        rows = h.local_rows()
        h.set_point(Y)
        f = h.cost()
        st = h.rtr(opts)
        Yall = h.get_point_all()
        h.close()
        return rows, f, st.cost, st.hessvecs, Yall

    res = run_ranks(N, one_rank)
    assert [q[0] for q in res][-2:] == [(18, 20), (21, 21)]
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_option("persist", 0)
    h.set_point(Y)
    f = h.cost(); st = h.rtr(opts); Y1 = h.get_point()
    h.close()
    for q in res:
        assert abs(q[1] - f) <= 1e-13 * abs(f) and abs(q[2] - st.cost) <= 1e-12 * abs(st.cost) and q[3] == st.hessvecs
        assert rel(q[4], Y1) < 1e-10


@pytest.mark.parametrize("halo", [0, 1])
@pytest.mark.parametrize("N", [2, 4])
def test_bench_entry_points_on_ranks(N, halo):
    """What bench.py --gpus N calls on every rank (snapshot / restore of the start point, whole RTR calls, the Hess-vec and
    tCG-trip timers) on N in-process ranks of the weak-scaled G81-family grid: same Hess-vec count on every rank and as the
    one-handle run of the same problem."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(20 * N, 50, seed=81)
    n, p = C.shape[0], 16
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=10, maxinner=30, tolgradnorm=1e-8)

    def session(h):
        h.set_point(Y0)
        h.point_snapshot()
        hv = 0
        for _ in range(2):
            h.point_restore()
            st = h.rtr(opts)
            hv += st.hessvecs
        h.set_point(Y0)
        ms, by, fl = h.bench_hessvec(5)
        trip = h.bench_tcg_trip(16)
        return hv, st.cost, ms > 0 and trip > 0

    def one_rank(r, group):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.comm_init_local(N, r, group)
        h.set_option("halo_exchange", halo)                # bench.py's second leg at N > 1
        q = session(h)
        h.close()
        return q

    res = run_ranks(N, one_rank)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_option("persist", 0)
    ref = session(h)
    h.close()
    for q in res:
        assert q[0] == ref[0] and q[2] and abs(q[1] - ref[1]) <= 1e-10 * abs(ref[1])


@pytest.mark.parametrize("N", [2, 4, 7])
@pytest.mark.parametrize("graph", ["grid", "random"])
def test_halo_exchange_is_bit_identical_to_all_gather(N, graph):
    """The rows a rank's rows of C reference arrive through the packed point-to-point exchange instead of the all-gather:
    the S*U kernels read the same numbers, so every result is bit-identical.  'random' is a graph without locality (every
    rank references rows of every other rank, the lists are long and ragged); N = 7 leaves an uneven last shard."""
    from manisdp_matlab_amd import _lib, problems, solvers
    _lib.load()
    if graph == "grid":
        C = problems.toroidal_grid_maxcut(36, 50, seed=2)
    else:
        rng = np.random.default_rng(8)
        n = 1500
        A = sp.random(n, n, density=4.0 / n, random_state=rng, data_rvs=lambda k: rng.choice([-1.0, 1.0], k))
        A = sp.triu(A, 1); A = A + A.T
        C = sp.csr_matrix(-0.25 * (sp.diags(np.asarray(A.sum(axis=1)).ravel()) - A))
    n, p = C.shape[0], 10
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    opts = _lib.default_opts(maxiter=6, maxinner=30, tolgradnorm=1e-9)
    Y0 = Y[:, :2] / np.linalg.norm(Y[:, :2], axis=1, keepdims=True)

    def runner(halo):
        def one_rank(r, group):
            h = _lib.Handle.onlyunitdiag(C, pcap=p)
            h.comm_init_local(N, r, group)
            h.set_option("halo_exchange", halo)
            h.set_point(Y)
            f = h.cost(); G = h.rgrad(); H = h.hessvec(U)
            st = h.rtr(opts)
            Yall = h.get_point_all()
            h.close()
            return f, G, H, st.cost, st.gradnorm, st.hessvecs, Yall
        return one_rank

    a = run_ranks(N, runner(0))
    b = run_ranks(N, runner(1))
    for qa, qb in zip(a, b):
        for x, y in zip(qa, qb):
            assert np.array_equal(np.asarray(x), np.asarray(y))

    def solve(halo):
        def one_rank(r, group):
            Ys, obj, d = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "tol": 1e-8, "comm": ("local", N, r, group), "halo_exchange": halo}, verbose=False)
            return obj, d["iters"], d["hessvecs"], d["status"], Ys
        return one_rank

    if N == 4:
        # whole solves: twenty outer iterations and 70 000 Hess-vecs amplify any last-bit difference of the replicated host
        # loops (N host threads share one BLAS pool here), so the end points are compared to rounding, not bit for bit
        sa, sb = run_ranks(N, solve(0)), run_ranks(N, solve(1))
        for qa, qb in zip(sa, sb):
            assert qa[3] == qb[3] and abs(qa[0] - qb[0]) <= 1e-8 * abs(qa[0])
            assert np.allclose(np.linalg.norm(qb[4], axis=1), 1.0, atol=1e-12)


@pytest.mark.parametrize("N", [2, 3])
def test_dense_shard_escape_matches_one_handle(N):
    """Pre-sharded dense C (BASELINE config 5's layout): the escape multiplies every rank's rows of C - diag(z) with the full
    Lanczos vector and all-gathers the pieces; the ranks must find the eigenpairs of the one-handle run and the
    same bits as each other."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    n, p, seed = 700, 6, 3
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=20, maxinner=40, tolgradnorm=1e-7)

    def session(h):
        h.set_point(Y)
        h.rtr(opts)
        lam, V, lmax, _ = h.escape_eigs(3, tol=1e-10, maxit=4000)
        return lam, V, lmax

    def one_rank(r, group):
        h = _lib.Handle.dense_synthetic(n, seed, nranks=N, rank=r, pcap=p)
        h.comm_init_local(N, r, group)
        q = session(h)
        h.close()
        return q

    res = run_ranks(N, one_rank)
    h = _lib.Handle.dense_synthetic(n, seed, pcap=p)
    lam, V, lmax = session(h)
    h.close()
    # against LAPACK on the explicit matrix at the one-handle point is test_gpu_dense's job; here: ranks == one handle
    for q in res:
        assert np.allclose(q[0], lam, rtol=0, atol=1e-8 * max(1.0, abs(lmax))) and abs(q[2] - lmax) <= 1e-6 * abs(lmax)
        for j in range(3):
            assert min(np.linalg.norm(q[1][:, j] - V[:, j]), np.linalg.norm(q[1][:, j] + V[:, j])) < 1e-5
        assert np.array_equal(q[0], res[0][0]) and np.array_equal(q[1], res[0][1])


@pytest.mark.parametrize("N", [2, 4])
def test_whole_solve_dense_synthetic_on_ranks(N):
    """ManiSDP_onlyunitdiag on the synthetic dense C of config 5 (scaled down), rows of C generated per rank on the device,
    sharded RTR, sharded escape product: against the one-handle solve and the oracle on the explicit matrix."""
    from manisdp_matlab_amd import problems, solvers
    from oracle import manisdp_ref as ref
    Csyn = problems.SyntheticDenseC(420, seed=7)
    n = Csyn.n
    rng = np.random.default_rng(2)
    Y0 = rng.standard_normal((n, 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)

    def one_rank(r, group):
        Y, obj, d = solvers.ManiSDP_onlyunitdiag(Csyn, {"Y0": Y0, "tol": 1e-8, "comm": ("local", N, r, group)}, verbose=False)
        return obj, d["status"], d["dinf"], Y

    res = run_ranks(N, one_rank)
    _, obj1, d1 = solvers.ManiSDP_onlyunitdiag(Csyn, {"Y0": Y0, "tol": 1e-8, "eig": "device"}, verbose=False)
    _, obj_h, d_h = solvers.ManiSDP_onlyunitdiag(Csyn, {"Y0": Y0, "tol": 1e-8}, verbose=False)     # host eig on Csyn.toarray()
    _, obj_o, d_o = ref.ManiSDP_onlyunitdiag(Csyn.toarray(), {"Y0": Y0, "tol": 1e-8})
    assert d1["status"] == 0 and d_h["status"] == 0 and d_o["status"] == 0
    assert abs(obj1 - obj_o) <= 1e-7 * abs(obj_o) and abs(obj_h - obj_o) <= 1e-7 * abs(obj_o)
    for q in res:
        assert q[1] == 0 and q[2] < 1e-8 and abs(q[0] - obj_o) <= 1e-7 * abs(obj_o)
        assert np.allclose(np.linalg.norm(q[3], axis=1), 1.0, atol=1e-12)
        assert q[0] == res[0][0] and np.array_equal(q[3], res[0][3])


@pytest.mark.parametrize("N", [2, 4])
def test_bench_multi_rank_legs_on_in_process_ranks(N):
    """bench.py's two extra legs of a --gpus N run -- the K steps once more with the halo exchange, the row-sharded dense
    config-5 shape -- with the communicator, the barrier and the max-over-ranks replaced by their in-process stand-ins."""
    import bench
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(20 * N, 50, seed=81)
    n, p = C.shape[0], 16
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=10, maxinner=30, tolgradnorm=1e-8)
    bar = threading.Barrier(N)
    box = [0.0] * N

    def one_rank(r, group):
        serial = [0]

        def join(h):
            serial[0] += 1
            h.comm_init_local(N, r, group * 10 + serial[0])

        def allmax(x):
            box[r] = x
            bar.wait()
            m = max(box)
            bar.wait()
            return m

        hl = bench.halo_leg(_lib, join, bar.wait, allmax, N, r, C, Y0, p, opts, 2, 1)
        k5 = bench.k5_dense_sharded(_lib, join, bar.wait, allmax, N, r, rows_per_gpu=300, p=16)
        return hl, k5

    res = run_ranks(N, one_rank)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_option("persist", 0)
    h.set_point(Y0); h.point_snapshot()
    hv = 0
    for _ in range(2):
        h.point_restore(); hv += h.rtr(opts).hessvecs
    h.close()
    for hl, k5 in res:
        assert hl["hessvecs"] == hv and hl["value"] > 0 and hl == res[0][0]
        assert k5["n"] == 300 * N and k5["S_times_X_products"] > 0 and k5["aggregate_TFLOPs_f64"] > 0 and k5 == res[0][1]


@pytest.mark.parametrize("halo", [0, 1])
@pytest.mark.parametrize("N", [1, 2, 4])
def test_sharded_trip_issues_one_exchange_and_one_allreduce(N, halo):
    """Collective calls per tCG trip of the row-sharded onlyunitdiag path with sparse C (msdp_trip1.hip): the exchange of the
    projected residual rows carries every rank's three sums of tCG.m:227,241, the product with the new direction follows by
    linearity, and only <mdelta, H mdelta> (tCG.m:166) takes an all-reduce of its own -- 2 calls per trip plus one exchange
    every 32nd trip (the direct refresh), against 3 with option trip1 = 0.  Same Hess-vec counts and stop reasons as the
    three-launch trip, end points equal to rounding, every rank holds the same bits."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(40, 50, seed=81)
    n, p = C.shape[0], 16
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=12, maxinner=80, tolgradnorm=1e-9)
    reps = 64                                                  # bench_tcg_trip: cost/grad (2 calls), tCG start (2 / 0), 2 + reps trips

    def one_rank(trip1):
        def body(r, group):
            h = _lib.Handle.onlyunitdiag(C, pcap=p)
            h.comm_init_local(N, r, group)
            h.set_option("halo_exchange", halo)
            h.set_option("trip1", trip1)
            h.set_option("xpersist", 0)                        # this test is about the chunked sharded trips (what RCCL ranks run)
            h.set_option("sweep", 2 if halo else 0)            # the windowed row traversal of the gather launch, on every rank's rows
            h.set_point(Y0)
            c0 = h.collective_calls()
            h.bench_tcg_trip(reps)
            calls = h.collective_calls() - c0
            h.set_point(Y0)
            st = h.rtr(opts)
            Yall = h.get_point_all()
            h.close()
            return calls, (st.hessvecs, st.accepted, st.rejected, st.iters), st.cost, Yall
        return body

    new = run_ranks(N, one_rank(1))
    old = run_ranks(N, one_rank(0))
    trips = reps + 2
    setup = old[0][0] - 3 * trips                                # cost / gradient evaluations in front of the tCG: the same on both paths
    assert 0 <= setup <= 6
    for q in old:
        assert q[0] == setup + 3 * trips                         # exchange, all-reduce (tCG.m:166), all-reduce (tCG.m:227,241)
    for q in new:
        assert q[0] == setup + 2 + 2 * trips + trips // 32       # first direct product, trips, refreshes
        assert q[1] == new[0][1] and np.array_equal(q[3], new[0][3])
    assert new[0][1] == old[0][1]
    assert abs(new[0][2] - old[0][2]) <= 1e-11 * abs(old[0][2])
    assert rel(new[0][3], old[0][3]) < 1e-8


@pytest.mark.parametrize("N,shape,p", [(2, (200, 200), 32), (2, (61, 50), 12), (4, (200, 200), 16), (3, (141, 142), 24), (2, (100, 200), 40)])
def test_cross_rank_persistent_tcg(N, shape, p):
    """ONE persistent tCG spanning the launches of N in-process ranks (k_tcg_persist_obl<..., XR>: 256 / N
    workgroups per rank, all co-resident; the grid reductions of tCG.m:166 and :227-241 run over shared uncached slots, the
    residual / direction rows travel through one shared exchange buffer) instead of lock-step chunks with an exchange and an
    all-reduce per trip.  Same Hess-vec counts, accept / reject sequence, stop codes and end point as ONE unsharded handle; the
    number of collective calls of a trustregions() call does not depend on the number of tCG trips (zero per trip); a bench call
    reports the trip time.  (2 x 20 000 rows at p = 32 is the shape the review names: 157 rows per workgroup, five row slots.)"""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(3)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=10, maxinner=60, tolgradnorm=1e-9)
    short = _lib.default_opts(maxiter=10, maxinner=7, tolgradnorm=1e-9)

    def body(r, group):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.comm_init_local(N, r, group)
        h.set_point(Y0)
        c0 = h.collective_calls()
        st = h.rtr(opts)
        c1 = h.collective_calls()
        path = h.tcg_path()
        Yall = h.get_point_all()
        h.set_point(Y0)
        c2 = h.collective_calls()
        st7 = h.rtr(short)
        c3 = h.collective_calls()
        h.set_point(Y0)
        trip_us = h.bench_tcg_trip(256) * 1e3
        h.close()
        return dict(path=path, stats=(st.hessvecs, st.accepted, st.rejected, st.iters, st.last_stop_inner), cost=st.cost, Y=Yall,
                    calls=c1 - c0, iters=st.iters, calls7=c3 - c2, iters7=st7.iters, hv7=st7.hessvecs, trip_us=trip_us)

    res = run_ranks(N, body)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y0)
    st = h.rtr(opts)
    Y1 = h.get_point()
    h.close()
    for q in res:
        assert q["path"] == 2                                      # the cross-rank persistent kernel ran
        assert q["stats"] == (st.hessvecs, st.accepted, st.rejected, st.iters, st.last_stop_inner)
        assert abs(q["cost"] - st.cost) <= 1e-10 * abs(st.cost)
        assert rel(q["Y"], Y1) < 1e-8
        assert np.array_equal(q["Y"], res[0]["Y"])
        # collectives per TR iteration only (retraction, cost / gradient at the proposal, decision): the same number per iteration
        # whether a tCG makes 7 trips or 60
        per_it = (q["calls"] - q["calls7"]) / max(q["iters"] - q["iters7"], 1) if q["iters"] != q["iters7"] else q["calls"] / max(q["iters"], 1)
        assert q["hv7"] < st.hessvecs
        assert abs(q["calls"] / max(q["iters"], 1) - q["calls7"] / max(q["iters7"], 1)) < 1.0 + 6.0 / max(q["iters7"], 1), (q["calls"], q["iters"], q["calls7"], q["iters7"], per_it)
    print("cross-rank persistent trip, N = %d, n = %d, p = %d: %.2f us" % (N, n, p, max(q["trip_us"] for q in res)))
    assert max(q["trip_us"] for q in res) < 40.0


def test_cross_rank_persistent_tcg_times_out_instead_of_hanging():
    """Workgroups that never arrive must not hang the launch: the bounded spins of the grid synchronisation turn the wait into an
    error word, the member that reads it breaks the group, and every member's trustregions() call returns MSDP_ECOMM.  Test hook
    debug_xr_skip on member 0: the next combined launch waits for eight workgroups more than it has."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(50, 60, seed=2)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(1)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)

    def body(r, group):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.comm_init_local(2, r, group)
        h.set_point(Y0)
        if r == 0:
            h.set_option("debug_xr_skip", 1)
        try:
            h.rtr(_lib.default_opts(maxiter=3, maxinner=10, tolgradnorm=1e-9))
            return "returned"
        except Exception as e:      # noqa: BLE001
            return "raised: %s" % e
        finally:
            h.close()

    out = run_ranks(2, body)
    assert all(o.startswith("raised") for o in out), out
