"""One member of a group of PROCESSES sharing a GPU (or owning one GPU each): tests/test_gpu_ipc_ranks.py starts N of these as fresh
processes.  argv: rank nranks shm_name rows cols p out.npz [device]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    rank, N, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rows, cols, p, out = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    if len(sys.argv) > 8:
        _lib.set_device(int(sys.argv[8]))
    C = problems.toroidal_grid_maxcut(rows, cols, seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(3)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=10, maxinner=60, tolgradnorm=1e-9)
    short = _lib.default_opts(maxiter=10, maxinner=7, tolgradnorm=1e-9)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.comm_init_ipc(N, rank, name)
    h.set_point(Y0)
    f0, G0 = h.cost(), h.rgrad()                                   # sharded operators through the staging slabs
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
    # a launch that waits for workgroups that do not exist: bounded spin -> MSDP_ECOMM on every member, no hang
    h.set_point(Y0)
    h.set_option("debug_xr_skip", 1)
    err = ""
    try:
        h.rtr(short)
    except _lib.MsdpError as e:
        err = str(e)
    r0, r1 = h.local_rows()
    np.savez(out, path=path, stats=np.array([st.hessvecs, st.accepted, st.rejected, st.iters, st.last_stop_inner]), cost=st.cost, Y=Yall,
             calls=c1 - c0, iters=st.iters, calls7=c3 - c2, iters7=st7.iters, hv7=st7.hessvecs, trip_us=trip_us, f0=f0, G0=G0, rows=np.array([r0, r1]),
             err=np.array(err))
    # (no h.close(): the group is broken after the provoked time-out; the process ends here)


if __name__ == "__main__":
    main()
