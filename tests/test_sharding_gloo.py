"""CPU test of the N > 1 data path with two gloo processes: the row partition, the uniform
all-gather of the thin direction before S*U and the all-reduce of the partial sums reproduce
the single-process Hess-vec of the oracle bit for bit in structure (allclose to 1e-14)."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from manisdp_matlab_amd import problems, sharding


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_rows, n_cols, p, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C = problems.toroidal_grid_maxcut(n_rows, n_cols, seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    r0, r1 = sharding.row_range(n, world, rank)
    Cl = sharding.shard_rows_csr(C, n, world, rank)
    Yl, Ul = Y[r0:r1], U[r0:r1]

    def allgather(local):
        slab = torch.from_numpy(sharding.pad_slab(local, n, world))
        outs = [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(outs, slab)
        return sharding.unpad_gathered([o.numpy() for o in outs], n)

    # cost state: eG rows need all of Y (ManiSDP_onlyunitdiag.m:118-119)
    Yfull = allgather(Yl)
    eG = np.sum((Cl @ Yfull) * Yl, axis=1, keepdims=True)
    # Hess-vec rows need all of U (:128-129); projections are row-local
    Ufull = allgather(Ul)
    eH = Cl @ Ufull
    Hl = eH - Yl * np.sum(Yl * eH, axis=1, keepdims=True) - Ul * eG
    part = torch.tensor([float(np.sum(Ul * Hl))], dtype=torch.float64)
    dist.all_reduce(part)
    Hfull = allgather(Hl)
    if rank == 0:
        np.save(out, np.concatenate([Hfull.ravel(), [part.item()]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,p", [((10, 13), 4), ((9, 7), 3)])
def test_two_rank_hessvec_matches_single_process(tmp_path, shape, p):
    from oracle import manisdp_ref as R
    out = str(tmp_path / "h.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, shape[0], shape[1], p, out), nprocs=2, join=True)
    got = np.load(out)
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    H = R.hessvec_onlyunitdiag(C, Y, U)
    assert np.allclose(got[:-1].reshape(n, p), H, rtol=0, atol=1e-13)
    assert abs(got[-1] - np.sum(U * H)) < 1e-11 * abs(np.sum(U * H))


def test_partition_properties():
    for n, w in [(20000, 8), (100000, 8), (17, 4), (5, 8), (64, 1)]:
        cap = sharding.row_capacity(n, w)
        ranges = [sharding.row_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert all(r1 - r0 <= cap for r0, r1 in ranges)


def _affine_worker(rank, world, port, out):
    """The row-sharded affine data path in NumPy over gloo: replicated operator state (A(.), Axb, the
    adjoint eS and AyU computed whole on every rank from the gathered factor), row-sharded dense contraction and row dots,
    all-reduced partial sums -- against the oracle's unsharded ManiSDP_unitdiag closures."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(2)
    d = 5
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
    e = rng.standard_normal(d)
    At, b, c, K = problems.bqpmom(d, Q, e)
    b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    c = np.asarray(c.todense()).ravel() if sp.issparse(c) else np.asarray(c, float).ravel()
    A = At.T.tocsr(); n, p, sigma = K["s"], 4, 0.7
    y = 0.1 * rng.standard_normal(b.size)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p)); U -= Y * np.sum(Y * U, axis=1, keepdims=True)
    r0, r1 = sharding.row_range(n, world, rank)
    Yl, Ul = Y[r0:r1], U[r0:r1]

    def allgather(local):
        slab = torch.from_numpy(sharding.pad_slab(local, n, world))
        outs = [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(outs, slab)
        return sharding.unpad_gathered([o.numpy() for o in outs], n)

    def allsum(v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t)
        return t.item()

    Yf = allgather(Yl)                                                   # yfull[slot]
    Axb = A @ (Yf @ Yf.T).ravel(order="F") - b - y / sigma               # replicated: identical on every rank
    eS = (c + sigma * (At @ Axb)).reshape((n, n), order="F")             # replicated adjoint
    CY = c.reshape((n, n), order="F")[r0:r1] @ Yf                         # my rows of C*Y
    cx = allsum(np.sum(CY * Yl))                                          # P_S1, all-reduced
    f = cx + 0.5 * sigma * float(Axb @ Axb)                               # P_AXB is replicated: NOT all-reduced
    eGl = 2.0 * (eS[r0:r1] @ Yf)                                          # my rows of the dense contraction
    YeG = np.sum(Yl * eGl, axis=1, keepdims=True)
    Gl = eGl - Yl * YeG
    Uf = allgather(Ul)                                                    # the direction, gathered before the Hess-vec
    AyU = (At @ (A @ (Yf @ Uf.T).ravel(order="F"))).reshape((n, n), order="F")   # replicated
    eHl = 2.0 * (eS[r0:r1] @ Uf) + 4.0 * sigma * (AyU[r0:r1] @ Yf)
    Hl = eHl - Yl * np.sum(Yl * eHl, axis=1, keepdims=True) - Ul * YeG
    dHd = allsum(np.sum(Ul * Hl))                                         # P_DHD, all-reduced
    Gf, Hf = allgather(Gl), allgather(Hl)
    if rank == 0:
        np.save(out, np.concatenate([[f, dHd], Gf.ravel(), Hf.ravel()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_affine_path_matches_oracle(tmp_path):
    from oracle import manisdp_ref as R
    out = str(tmp_path / "a.npy")
    mp.spawn(_affine_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    rng = np.random.default_rng(2)
    d = 5
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
    e = rng.standard_normal(d)
    At, b, c, K = problems.bqpmom(d, Q, e)
    b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    c = np.asarray(c.todense()).ravel() if sp.issparse(c) else np.asarray(c, float).ravel()
    n, p, sigma = K["s"], 4, 0.7
    y = 0.1 * rng.standard_normal(b.size)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p)); U -= Y * np.sum(Y * U, axis=1, keepdims=True)
    prob = R._UnitDiagProblem(At, b, c, n, p)
    prob.y, prob.sigma = y, sigma
    f = prob.cost(Y); G = prob.grad(Y); H = prob.hess(Y, U)
    assert abs(got[0] - f) <= 1e-12 * max(1.0, abs(f))
    assert abs(got[1] - np.sum(U * H)) <= 1e-11 * max(1.0, abs(np.sum(U * H)))
    assert np.allclose(got[2:2 + n * p].reshape(n, p), G, rtol=0, atol=1e-12 * np.linalg.norm(G))
    assert np.allclose(got[2 + n * p:].reshape(n, p), H, rtol=0, atol=1e-12 * np.linalg.norm(H))


def _halo_worker(rank, world, port, kind, p, out):
    """The halo exchange of option halo_exchange with gloo point-to-point messages: pack the rows each peer's rows of C
    reference, exchange, scatter to global positions, multiply -- only referenced rows of the gather buffer are defined."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C = _halo_matrix(kind)
    n = C.shape[0]
    rng = np.random.default_rng(1)
    U = rng.standard_normal((n, p))
    r0, r1 = sharding.row_range(n, world, rank)
    Cl = sharding.shard_rows_csr(C, n, world, rank)
    send, recv = sharding.halo_lists(C, world, rank)
    full = np.full((n, p), np.nan)                       # rows nobody sends stay undefined: they must never be read
    full[r0:r1] = U[r0:r1]
    reqs, bufs = [], {}
    for q in range(world):
        if q == rank:
            continue
        if send[q].size:
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(U[r0:r1][send[q]])), dst=q))
        if recv[q].size:
            bufs[q] = torch.empty((recv[q].size, p), dtype=torch.float64)
            reqs.append(dist.irecv(bufs[q], src=q))
    for r in reqs:
        r.wait()
    for q, t in bufs.items():
        full[recv[q]] = t.numpy()
    ref = np.unique(Cl.indices)
    assert not np.isnan(full[ref]).any()                 # every referenced row arrived
    Hl = Cl @ np.nan_to_num(full)                        # (the product touches referenced rows only)
    nrecv = sum(v.size for v in recv)
    slab = torch.from_numpy(sharding.pad_slab(Hl, n, world))
    outs = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(outs, slab)
    cnt = torch.tensor([float(nrecv)], dtype=torch.float64)
    dist.all_reduce(cnt)
    if rank == 0:
        np.save(out, np.concatenate([sharding.unpad_gathered([o.numpy() for o in outs], n).ravel(), [cnt.item()]]))
    dist.barrier()
    dist.destroy_process_group()


def _halo_matrix(kind):
    if kind == "grid":
        return problems.toroidal_grid_maxcut(12, 9, seed=7)
    rng = np.random.default_rng(8)
    n = 90
    A = sp.random(n, n, density=5.0 / n, random_state=rng, data_rvs=lambda k: rng.choice([-1.0, 1.0], k))
    A = sp.triu(A, 1); A = A + A.T
    return sp.csr_matrix(-0.25 * (sp.diags(np.asarray(A.sum(axis=1)).ravel()) - A))


@pytest.mark.parametrize("kind,world", [("grid", 2), ("grid", 3), ("random", 3)])
def test_halo_exchange_protocol(tmp_path, kind, world):
    """sharding.halo_lists (the host-side statement of halo_setup in msdp_api.hip): the lists of the two ends of every pair
    agree, the exchanged rows are exactly the referenced ones, and the local products equal the rows of C*U; on the grid a
    rank receives two grid rows from each neighbour, not the whole direction."""
    out = str(tmp_path / "halo.npy")
    mp.spawn(_halo_worker, args=(world, _free_port(), kind, 3, out), nprocs=world, join=True)
    got = np.load(out)
    C = _halo_matrix(kind)
    n = C.shape[0]
    U = np.random.default_rng(1).standard_normal((n, 3))
    assert np.allclose(got[:-1].reshape(n, 3), C @ U, rtol=0, atol=1e-13)
    lists = [sharding.halo_lists(C, world, r) for r in range(world)]
    for a in range(world):
        for bq in range(world):
            if a != bq:
                a0, _ = sharding.row_range(n, world, a)
                assert np.array_equal(lists[a][0][bq] + a0, lists[bq][1][a])        # what a sends to b is what b expects from a
    if kind == "grid":
        assert got[-1] < 0.5 * world * n                 # far fewer rows travel than an all-gather moves ((world-1) n per rank)


def _trip1_start(C, Y, warm):
    """`warm` trust-region iterations of the oracle from Y: near a stationary point a tCG runs its whole budget (at a random
    point it leaves through negative curvature after a trip or two)."""
    if not warm:
        return Y
    from oracle import manisdp_ref as R, manopt_rtr
    prob = R._OnlyUnitDiagProblem(C, C.shape[0], Y.shape[1], q1="correct")
    Y2, _, _ = manopt_rtr.trustregions(prob, Y.copy(), warm, 50, 1e-9)
    return Y2


def _trip1_worker(rank, world, port, shape, p, maxinner, warm, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    Y = _trip1_start(C, Y, warm)
    r0, r1 = sharding.row_range(n, world, rank)
    Cl = sharding.shard_rows_csr(C, n, world, rank)
    calls = {"exchange": 0, "allreduce": 0}

    def allgather(local):
        slab = torch.from_numpy(sharding.pad_slab(local, n, world))
        outs = [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(outs, slab)
        return sharding.unpad_gathered([o.numpy() for o in outs], n)

    def exchange(rows, sums):
        # rows and sums in one message per rank: the sums ride in a pad row of the slab
        calls["exchange"] += 1
        tail = np.zeros((1, p)); tail[0, :3] = sums
        t = torch.from_numpy(np.vstack([sharding.pad_slab(rows, n, world), tail]))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        full = sharding.unpad_gathered([o.numpy()[:-1] for o in outs], n)
        return full, [list(o.numpy()[-1, :3]) for o in outs]

    def allreduce(x):
        calls["allreduce"] += 1
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    Yl = Y[r0:r1]
    YC = Cl @ Y
    eGl = np.sum(YC * Yl, axis=1, keepdims=True)              # ManiSDP_onlyunitdiag.m:118-119
    gl = YC - Yl * eGl                                         # :123-124
    eta, Heta, j, stop = sharding.tcg_one_allreduce(Cl, Yl, gl, eGl, 1e3, maxinner, exchange, allreduce)
    E = allgather(eta); H = allgather(Heta)
    if rank == 0:
        np.save(out, np.concatenate([E.ravel(), H.ravel(), [j, stop, calls["exchange"], calls["allreduce"]]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("maxinner,warm", [(40, 0), (40, 10), (5, 15), (80, 15)])
def test_tcg_with_one_exchange_and_one_allreduce_per_trip(tmp_path, world, maxinner, warm):
    """The protocol of the row-sharded tCG trip (csrc/msdp_trip1.hip; host statement: sharding.tcg_one_allreduce) between
    gloo processes: the exchange of the projected residual rows carries every rank's partial sums, the product with the new
    direction follows by linearity, only <mdelta, H mdelta> is all-reduced.  Same step, same trip count and stop code as the
    oracle's tCG (oracle/manopt_rtr.py, tCG.m:95-292); 1 exchange + 1 all-reduce per trip plus the direct refresh every 32nd."""
    from oracle import manisdp_ref as R, manopt_rtr
    shape, p = (12, 11), 5
    out = str(tmp_path / "t.npy")
    mp.spawn(_trip1_worker, args=(world, _free_port(), shape, p, maxinner, warm, out), nprocs=world, join=True)
    got = np.load(out)
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    Y = _trip1_start(C, Y, warm)
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    prob.cost(Y)
    g = prob.grad(Y)
    eta, Heta, j, stop = manopt_rtr.tCG(prob, Y, g, 1e3, maxinner)
    E, H = got[:n * p].reshape(n, p), got[n * p:2 * n * p].reshape(n, p)
    jj, sstop, nex, nar = (int(v) for v in got[2 * n * p:])
    assert (jj, sstop) == (j, stop)
    assert np.linalg.norm(E - eta) <= 1e-10 * np.linalg.norm(eta)
    # Heta is a sum of alpha*H*mdelta terms that nearly cancels near a stationary point: rounding is relative to |C| |eta|
    normC = float(abs(C).sum(axis=1).max())
    assert np.linalg.norm(H - Heta) <= 1e-10 * np.linalg.norm(Heta) + 1e-11 * normC * np.linalg.norm(eta)
    if warm == 15:
        assert (j, stop) == (maxinner, 5)                        # the whole budget: two refreshes at maxinner = 80
    cont = j - 1                                                 # trips that went on to a next one
    reached = cont if stop in (1, 2) else j                      # negative curvature / boundary (tCG.m:183) leave before the exchange
    assert nex == 1 + reached + cont // 32                       # gradient rows, one per trip, the direct refresh every 32nd
    assert nar == 2 + cont                                       # |grad|^2, the first product, one per further trip
