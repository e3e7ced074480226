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
    if os.environ.get("MSDP_TEST_XR_TWOLEVEL"):                    # the two-level grid reductions also where the flat ones would do
        h.set_option("xr_twolevel", 1)
    h.set_point(Y0)
    f0, G0 = h.cost(), h.rgrad()                                   # sharded operators through the staging slabs
    c0 = h.collective_calls()
    import time
    t0 = time.perf_counter()
    st = h.rtr(opts)
    rtr_us_per_hv = (time.perf_counter() - t0) * 1e6 / max(st.hessvecs, 1)
    c1 = h.collective_calls()
    path = h.tcg_path()
    Yall = h.get_point_all()
    # the same call with the rest of every TR iteration on the sharded kernels and their collectives (option xtail = 0)
    light = bool(os.environ.get("MSDP_TEST_LIGHT"))                # eight processes on one device: the hardware queues are time-sliced,
    if light:                                                      # every call costs seconds -- the variants are covered at N <= 4
        stc, rtr_us_per_hv_coll, calls_coll, Ycoll = st, rtr_us_per_hv, 3 * st.iters, Yall
    else:
        h.set_option("xtail", 0)
        h.set_point(Y0)
        cc0 = h.collective_calls()
        t0 = time.perf_counter()
        stc = h.rtr(opts)
        rtr_us_per_hv_coll = (time.perf_counter() - t0) * 1e6 / max(stc.hessvecs, 1)
        calls_coll = h.collective_calls() - cc0
        Ycoll = h.get_point_all()
        h.set_option("xtail", 1)
    h.set_point(Y0)
    c2 = h.collective_calls()
    st7 = h.rtr(short)
    c3 = h.collective_calls()
    h.set_point(Y0)
    trip_us = h.bench_tcg_trip(64 if light else 256) * 1e3
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
             err=np.array(err), rtr_us_per_hv=rtr_us_per_hv, rtr_us_per_hv_coll=rtr_us_per_hv_coll, calls_coll=calls_coll, Ycoll=Ycoll,
             stats_coll=np.array([stc.hessvecs, stc.accepted, stc.rejected, stc.iters, stc.last_stop_inner]))
    # (no h.close(): the group is broken after the provoked time-out; the process ends here)


if __name__ == "__main__":
    main()
