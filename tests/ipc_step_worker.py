"""One member of a group of PROCESSES running bench.py's step on G81 (tests/test_gpu_benchmarked_step.py): the seed-0 start point, one
trustregions() call with maxiter = 40, maxinner = 100.  argv: rank nranks shm_name p out.npz [device]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    rank, N, name, p, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    if len(sys.argv) > 6:
        _lib.set_device(int(sys.argv[6]))
    C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.comm_init_ipc(N, rank, name)
    h.set_point(Y0)
    st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
    path = h.tcg_path()
    Yall = h.get_point_all()
    np.savez(out, path=path, stats=np.array([st.iters, st.hessvecs, st.accepted, st.rejected, st.last_stop_inner]), cost=st.cost,
             gradnorm=st.gradnorm, Y=Yall)
    h.close()


if __name__ == "__main__":
    main()
