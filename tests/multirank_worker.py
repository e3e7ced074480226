"""Child process of tests/test_gpu_comm.py::test_two_ranks_*: rank RANK of WORLD_SIZE on GPU LOCAL_RANK.  torch (its
bundled HIP runtime + RCCL) comes up BEFORE libmanisdp_hip.so; the 128-byte RCCL id travels through
torch.distributed; every rank runs the same calls on its row shard and rank 0 writes the combined result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    torch.cuda.init()
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", rank))
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    _lib.set_device(rank)
    uid = [_lib.Handle.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    rng = np.random.default_rng(11)
    if case == "affine":
        return affine_case(out, rank, world, uid[0], dist, torch, _lib, problems)
    if case in ("solve", "solve-dense"):
        return solve_case(out, rank, world, uid[0], dist, _lib, problems, dense=(case == "solve-dense"))
    if case in ("sparse", "sparse-halo"):
        C = problems.toroidal_grid_maxcut(61, 50, seed=4)           # n = 3050: ragged last shard for N = 4, 8
        n, p = C.shape[0], 12
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
    else:
        n, p = 1000, 24
        h = _lib.Handle.dense_synthetic(n, 3, nranks=world, rank=rank, pcap=p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h.comm_init(world, rank, uid[0])
    if case == "sparse-halo":
        h.set_option("halo_exchange", 1)
    r0, r1 = h.local_rows()
    h.set_point(Y)
    f = h.cost()
    G = h.rgrad()
    H = h.hessvec(U)
    st = h.rtr(_lib.default_opts(maxiter=8, maxinner=25, tolgradnorm=1e-9))
    Yout = h.get_point()
    z = h.get_z()
    h.close()
    # every rank filled its own rows only (the others are zero): sum over the ranks = the full arrays
    parts = [torch.from_numpy(a).cuda() for a in (G, H, Yout, z)]
    for t in parts:
        dist.all_reduce(t)
    if rank == 0:
        np.savez(out, f=f, G=parts[0].cpu().numpy(), H=parts[1].cpu().numpy(), Y=parts[2].cpu().numpy(), z=parts[3].cpu().numpy(),
                 cost=st.cost, gradnorm=st.gradnorm, hessvecs=st.hessvecs, accepted=st.accepted, rejected=st.rejected,
                 rows=np.array([r0, r1]))
    dist.barrier()
    dist.destroy_process_group()


def affine_case(out, rank, world, uid, dist, torch, _lib, problems):
    """BQP d = 10 moment relaxation (ManiSDP_unitdiag) with its rows sharded: operators, one trustregions() call, the AL
    bookkeeping and the replicated escape; then a whole solve through the host loop with options['comm']."""
    import scipy.sparse as sp
    from manisdp_matlab_amd import solvers
    gold = os.path.join(ROOT, "tests", "golden")
    Q = np.loadtxt(os.path.join(gold, "bqp_Q_10_1.txt.gz"), delimiter=",")
    e = np.loadtxt(os.path.join(gold, "bqp_e_10_1.txt.gz"), delimiter=",")
    At, b, c, K = problems.bqpmom(10, Q, e)
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    At = sp.csc_matrix(At); At.sort_indices()
    n, m, p = K["s"], b.size, 6
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = 0.3 * rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(m)
    h = _lib.Handle.affine(_lib.KIND_UNITDIAG, At, b, c, n)
    h.comm_init(world, rank, uid)
    h.set_multipliers(y, 0.7)
    h.set_point(Y)
    f = h.cost(); G = h.rgrad(); H = h.hessvec(h.proj(U))
    co = h.linesearch_cost(U, 0.5)
    st = h.rtr(_lib.default_opts(maxiter=3, maxinner=15, tolgradnorm=1e-8))
    Yr = h.get_point_all()
    obj, Ax = h.al_primal(m)
    z = h.al_dual(y)
    lam, V, lmax, _ = h.escape_eigs_dual(3, tol=1e-10, maxit=4000)
    h.close()
    parts = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (G, H)]
    for t in parts:
        dist.all_reduce(t)                      # every rank filled its own rows only
    # the replicated quantities must be identical on every rank
    same = torch.from_numpy(np.concatenate([[f, co, st.cost, obj, lmax], Ax, z, lam, Yr.ravel()])).cuda()
    lo, hi = same.clone(), same.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    replicated_ok = bool(torch.equal(lo, hi))
    rng0 = np.random.default_rng(5)
    Y0 = rng0.standard_normal((n, 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    Ys, objs, ds = solvers.ManiSDP_unitdiag(At, b, c, K, {"Y0": Y0, "tol": 1e-8, "comm": (world, rank, _second_uid(dist, _lib, rank))}, verbose=False)
    if rank == 0:
        np.savez(out, f=f, G=parts[0].cpu().numpy(), H=parts[1].cpu().numpy(), co=co, cost=st.cost, hessvecs=st.hessvecs, Y=Yr, obj=obj,
                 Ax=Ax, z=z, lam=lam, lmax=lmax, replicated_ok=replicated_ok, solve_obj=objs, solve_status=ds["status"],
                 solve_eta=max(ds["gap"], ds["pinf"], ds["dinf"]))
    dist.barrier()
    dist.destroy_process_group()


def solve_case(out, rank, world, uid, dist, _lib, problems, dense=False):
    """ManiSDP_onlyunitdiag end to end on a row-sharded toroidal-grid MaxCut problem (sparse C): sharded RTR, replicated
    escape, replicated host loop; dense: the synthetic dense C generated per rank, sharded escape product."""
    from manisdp_matlab_amd import solvers
    C = problems.SyntheticDenseC(1000, seed=6) if dense else problems.toroidal_grid_maxcut(40, 50, seed=6)
    rng = np.random.default_rng(9)
    Y0 = rng.standard_normal((C.shape[0], 2)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0, "tol": 1e-8, "comm": (world, rank, uid)}, verbose=False)
    if rank == 0:
        np.savez(out, obj=obj, status=data["status"], dinf=data["dinf"], iters=data["iters"], Y=Y, z=data["z"])
    dist.barrier()
    dist.destroy_process_group()


def _second_uid(dist, _lib, rank):
    uid = [_lib.Handle.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    return uid[0]


if __name__ == "__main__":
    main()
