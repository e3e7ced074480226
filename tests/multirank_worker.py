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
    if case == "sparse":
        C = problems.toroidal_grid_maxcut(61, 50, seed=4)           # n = 3050: ragged last shard for N = 4, 8
        n, p = C.shape[0], 12
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
    else:
        n, p = 1000, 24
        h = _lib.Handle.dense_synthetic(n, 3, nranks=world, rank=rank, pcap=p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h.comm_init(world, rank, uid[0])
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


if __name__ == "__main__":
    main()
