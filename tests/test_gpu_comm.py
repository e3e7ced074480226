"""GPU test of the RCCL data path on ONE GPU: a size-1 communicator routes every exchange through the same
ncclAllGather / ncclAllReduce calls the 8-GPU run uses (all-gather of the thin direction into the gather
buffer before S*U, all-reduce of the partial-sum arrays, chunked tCG without host-timing dependence).
Results must be bit-identical to the communicator-free path."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu


def test_size1_communicator_is_bit_identical(monkeypatch):
    monkeypatch.setenv("MSDP_NO_PERSIST", "1")       # compare with the communicator-free CHUNKED path (same kernels)
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n, p = C.shape[0], 12
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    res = []
    for use_comm, trip1 in ((False, 0), (True, 0), (True, 1)):
        h = _lib.Handle.onlyunitdiag(C)
        if use_comm:
            h.comm_init(1, 0, _lib.Handle.comm_unique_id())
            assert h.local_rows() == (0, n)
        h.set_option("trip1", trip1)                      # 0: the three-launch trip on both sides, the same kernels
        h.set_point(Y)
        H = h.hessvec(U)
        G = h.rgrad()
        st = h.rtr(_lib.default_opts(maxiter=10, maxinner=30, tolgradnorm=1e-8))
        res.append((H, G, st.cost, st.gradnorm, st.hessvecs, st.accepted, h.get_point()))
        h.close()
    a, b, c = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and a[5] == b[5]
    assert np.array_equal(a[6], b[6])
    # the default sharded trip (msdp_trip1.hip: one exchange with the sums riding along through a grouped ncclAllGather, one
    # all-reduce) forms C*mdelta by linearity: same decisions, results equal to rounding
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    assert a[4] == c[4] and a[5] == c[5]
    assert abs(a[2] - c[2]) <= 1e-12 * abs(a[2]) and abs(a[3] - c[3]) <= 1e-9 * abs(a[3])
    assert np.linalg.norm(a[6] - c[6]) <= 1e-9 * np.linalg.norm(a[6])


def test_rccl_point_to_point_calls_of_the_halo_exchange():
    """ncclGroupStart .. ncclSend + ncclRecv .. ncclGroupEnd on the handle's communicator and stream, the handle's own rank
    as peer (the one pairing a single GPU offers): the RCCL half of the halo exchange; its lists are covered by
    tests/test_gpu_local_ranks.py."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(10, 10, seed=1)
    h = _lib.Handle.onlyunitdiag(C)
    h.comm_init(1, 0, _lib.Handle.comm_unique_id())
    rng = np.random.default_rng(2)
    for count in (1, 400 * 32, 1 << 20):
        x = rng.standard_normal(count)
        assert np.array_equal(h.debug_p2p_self(x), x)
    h.close()


@pytest.mark.parametrize("N", [2, 3, 8])
@pytest.mark.parametrize("case,p", [("G11", 12), ("G1", 7)])
def test_sparse_shards_match_unsharded(N, case, p):
    """Every row shard of the sparse path (rank r of N standing alone on one GPU; the test fills the gather buffer
    that the RCCL all-gather fills in the N-GPU run) reproduces its rows of the cost state, the gradient and the
    Hess-vec of the unsharded problem: local CSR/ELL rows with global column indices, row0 != 0, ragged last shard."""
    from manisdp_matlab_amd import _lib, problems
    from oracle import manisdp_ref as R
    _lib.load()
    C = problems.maxcut_cost_matrix(golden_path(case + ".txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(N)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    z_ref = np.asarray(np.sum((C @ Y) * Y, axis=1)).ravel()
    covered = np.zeros(n, bool)
    for r in range(N):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.debug_shard(N, r)
        r0, r1 = h.local_rows()
        covered[r0:r1] = True
        h.set_point(Y)
        h.debug_set_full_rows(Y)
        G = h.rgrad()
        assert np.linalg.norm(G[r0:r1] - G_ref[r0:r1]) <= 1e-12 * np.linalg.norm(G_ref)
        assert np.linalg.norm(h.get_z()[r0:r1] - z_ref[r0:r1]) <= 1e-12 * np.linalg.norm(z_ref)
        h.debug_set_full_rows(U)
        H = h.hessvec(U)
        assert np.linalg.norm(H[r0:r1] - H_ref[r0:r1]) <= 1e-12 * np.linalg.norm(H_ref)
        assert h.tcg_path() == 0                      # shards never take the single-GPU persistent path
        h.close()
    assert covered.all()


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("case", ["sparse", "dense", "sparse-halo"])
@pytest.mark.parametrize("N", [2, 4, 8])
def test_ranks_on_separate_gpus_match_one_rank(tmp_path, case, N):
    """The real thing: N processes, one per GPU, RCCL all-gather of the direction before every S*U and all-reduce of the
    partial sums, against the same calls on one unsharded handle.  Needs N visible GPUs: skipped
    on the single-GPU box, runs wherever the driver has a multi-GPU node."""
    import os, subprocess, sys
    import torch
    if torch.cuda.device_count() < N:
        pytest.skip("needs %d GPUs, %d visible" % (N, torch.cuda.device_count()))
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    out = str(tmp_path / "ranks.npz")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multirank_worker.py")
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, case, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    rng = np.random.default_rng(11)
    if case.startswith("sparse"):                     # "sparse-halo": option halo_exchange (grouped ncclSend / ncclRecv)
        C = problems.toroidal_grid_maxcut(61, 50, seed=4)
        n, p = C.shape[0], 12
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
    else:
        n, p = 1000, 24
        h = _lib.Handle.dense_synthetic(n, 3, pcap=p)
    h.set_option("persist", 0)                        # the sharded run uses the chunked kernels
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h.set_point(Y)
    f = h.cost(); G = h.rgrad(); H = h.hessvec(U)
    st = h.rtr(_lib.default_opts(maxiter=8, maxinner=25, tolgradnorm=1e-9))
    Yout = h.get_point(); z = h.get_z()
    h.close()
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)   # noqa: E731
    assert abs(float(got["f"]) - f) <= 1e-12 * abs(f)
    assert rel(got["G"], G) < 1e-13 and rel(got["H"], H) < 1e-13
    # the partial sums are grouped differently (per shard, then all-reduced), so the solve agrees to rounding
    assert int(got["hessvecs"]) == st.hessvecs and int(got["accepted"]) == st.accepted and int(got["rejected"]) == st.rejected
    assert abs(float(got["cost"]) - st.cost) <= 1e-10 * abs(st.cost)
    assert rel(got["Y"], Yout) < 1e-8 and rel(got["z"], z) < 1e-8


@pytest.mark.parametrize("N", [2, 4])
def test_affine_ranks_on_separate_gpus_match_one_rank(tmp_path, N):
    """Row-sharded ManiSDP_unitdiag (BQP d = 10) on N GPUs against one unsharded handle: operators, trustregions(), AL
    bookkeeping, the replicated escape, a whole solve.  Needs N visible GPUs (skipped on the single-GPU box; the
    single-GPU stand-ins are in tests/test_gpu_affine_sharded.py)."""
    import os, subprocess, sys
    import scipy.sparse as sp
    import torch
    if torch.cuda.device_count() < N:
        pytest.skip("needs %d GPUs, %d visible" % (N, torch.cuda.device_count()))
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    out = str(tmp_path / "ranks.npz")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multirank_worker.py")
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, "affine", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
    e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(10, Q, e)
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    At = sp.csc_matrix(At); At.sort_indices()
    n, m, p = K["s"], b.size, 6
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = 0.3 * rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(m)
    h = _lib.Handle.affine(_lib.KIND_UNITDIAG, At, b, c, n)
    h.set_multipliers(y, 0.7)
    h.set_point(Y)
    f = h.cost(); G = h.rgrad(); H = h.hessvec(h.proj(U))
    co = h.linesearch_cost(U, 0.5)
    st = h.rtr(_lib.default_opts(maxiter=3, maxinner=15, tolgradnorm=1e-8))
    Yr = h.get_point()
    obj, Ax = h.al_primal(m)
    z = h.al_dual(y)
    lam, V, lmax, _ = h.escape_eigs_dual(3, tol=1e-10, maxit=4000)
    h.close()
    rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)   # noqa: E731
    assert bool(got["replicated_ok"])
    assert abs(float(got["f"]) - f) <= 1e-12 * abs(f) and abs(float(got["co"]) - co) <= 1e-12 * abs(co)
    assert rel(got["G"], G) < 1e-12 and rel(got["H"], H) < 1e-12
    assert int(got["hessvecs"]) == st.hessvecs and abs(float(got["cost"]) - st.cost) <= 1e-10 * abs(st.cost)
    assert rel(got["Y"], Yr) < 1e-8 and rel(got["Ax"], Ax) < 1e-8 and rel(got["z"], z) < 1e-8
    assert abs(float(got["obj"]) - obj) <= 1e-9 * abs(obj)
    assert rel(got["lam"][np.isfinite(got["lam"])], lam[np.isfinite(lam)]) < 1e-6 and abs(float(got["lmax"]) - lmax) <= 1e-6 * abs(lmax)
    from manisdp_matlab_amd import solvers
    rng0 = np.random.default_rng(5)
    Y0 = rng0.standard_normal((n, 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    _, obj1, d1 = solvers.ManiSDP_unitdiag(At, b, c, K, {"Y0": Y0, "tol": 1e-8}, verbose=False)
    assert int(got["solve_status"]) == 0 and float(got["solve_eta"]) < 1e-8
    assert abs(float(got["solve_obj"]) - obj1) <= 1e-6 * max(1.0, abs(obj1))


def test_size1_communicator_onlyunitdiag_solve(monkeypatch):
    """ManiSDP_onlyunitdiag through the host loop with options['comm'] (size-1 communicator): sharded RTR, the escape on the
    replicated copy of C with gathered z and Y, get_point_all / get_z_all, the device-resident factor -- against the
    communicator-free solve of the same instance (G1; KKT 1e-8, same optimum)."""
    from manisdp_matlab_amd import _lib, problems, solvers
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    rng = np.random.default_rng(2)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    o = {"Y0": Y0, "tol": 1e-8, "eig": "device"}
    Ya, obja, da = solvers.ManiSDP_onlyunitdiag(C, dict(o), verbose=False)
    Yb, objb, db = solvers.ManiSDP_onlyunitdiag(C, dict(o, comm=(1, 0, _lib.Handle.comm_unique_id())), verbose=False)
    assert da["status"] == 0 and db["status"] == 0 and db["dinf"] < 1e-8
    assert abs(obja - objb) <= 1e-7 * abs(obja)
    assert np.allclose(np.linalg.norm(Yb, axis=1), 1.0, atol=1e-12) and db["z"].shape == (C.shape[0],)


def test_size1_communicator_escape_uses_the_persistent_lanczos_kernels():
    """The replicated escape of a row-sharded sparse handle is a single-GPU computation on every rank: it runs the persistent
    Lanczos kernels on the replicated CSR copy, and finds bit for bit what the communicator-free handle finds (undeflated
    and deflated runs, four pairs)."""
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(60, 100, seed=9)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = []
    for use_comm in (False, True):
        h = _lib.Handle.onlyunitdiag(C)
        if use_comm:
            h.comm_init(1, 0, _lib.Handle.comm_unique_id())
        h.set_option("escape_warm", 0)
        h.set_point(Y)
        h.cost()
        import time
        t0 = time.perf_counter()
        lam, V, lmax, steps = h.escape_eigs(4, tol=1e-10, maxit=20000)
        out.append((lam, V, lmax, steps, time.perf_counter() - t0))
        h.close()
    a, b = out
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert b[4] < 3.0 * a[4] + 0.05          # the multi-kernel path (seven launches per step) is ~10x slower at this size


def test_size1_communicator_dense_synthetic_solve():
    """The same for the pre-sharded synthetic dense C (config 5's layout, scaled down): sharded RTR and the escape with the
    sharded product + ncclAllGather of the pieces per Lanczos step, against the communicator-free solve."""
    from manisdp_matlab_amd import _lib, problems, solvers
    Csyn = problems.SyntheticDenseC(600, seed=5)
    rng = np.random.default_rng(4)
    Y0 = rng.standard_normal((Csyn.n, 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    o = {"Y0": Y0, "tol": 1e-8, "eig": "device"}
    Ya, obja, da = solvers.ManiSDP_onlyunitdiag(Csyn, dict(o), verbose=False)
    Yb, objb, db = solvers.ManiSDP_onlyunitdiag(Csyn, dict(o, comm=(1, 0, _lib.Handle.comm_unique_id())), verbose=False)
    assert da["status"] == 0 and db["status"] == 0 and db["dinf"] < 1e-8
    assert abs(obja - objb) <= 1e-7 * abs(obja)
    # LAPACK on the explicit matrix at the returned point: the reported dinf is the true one
    z = db["z"]
    w = np.linalg.eigvalsh(Csyn.toarray() - np.diag(z))
    assert max(0.0, -w[0]) / (1.0 + w[-1]) < 1e-8 and abs(np.sum(z) - objb) <= 1e-10 * abs(objb)


@pytest.mark.parametrize("case", ["solve", "solve-dense"])
@pytest.mark.parametrize("N", [2, 4, 8])
def test_onlyunitdiag_solve_on_separate_gpus(tmp_path, N, case):
    """Whole ManiSDP_onlyunitdiag solve on N GPUs (sharded RTR, replicated escape and host loop) against the one-GPU solve;
    'solve-dense': the pre-sharded synthetic dense C with the sharded escape product.
    Needs N visible GPUs (skipped on the single-GPU box; its single-GPU stand-in is test_size1_communicator_onlyunitdiag_solve)."""
    import os, subprocess, sys
    import torch
    if torch.cuda.device_count() < N:
        pytest.skip("needs %d GPUs, %d visible" % (N, torch.cuda.device_count()))
    from manisdp_matlab_amd import problems, solvers
    out = str(tmp_path / "solve.npz")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multirank_worker.py")
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, case, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    C = problems.toroidal_grid_maxcut(40, 50, seed=6) if case == "solve" else problems.SyntheticDenseC(1000, seed=6)
    rng = np.random.default_rng(9)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    _, obj1, d1 = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "tol": 1e-8, "eig": "device"}, verbose=False)
    assert int(got["status"]) == 0 and float(got["dinf"]) < 1e-8 and d1["status"] == 0
    assert abs(float(got["obj"]) - obj1) <= 1e-7 * abs(obj1)
    assert np.allclose(np.linalg.norm(got["Y"], axis=1), 1.0, atol=1e-12)
