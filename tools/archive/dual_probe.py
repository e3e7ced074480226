"""Dual approach (ManiDSDP_unitdiag) on the SOS relaxation of a random BQP with d variables, as
example/dual/example_bqp_dual.m builds it: argv = d [d ...] [--eig=host|device]  (GPU path against the CPU restatement: tests/test_gpu_dual.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
args = sys.argv[1:]
eig = None
for a in list(args):
    if a.startswith("--eig="):
        eig = a.split("=", 1)[1]; args.remove(a)
for d in [int(a) for a in args]:
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
    e = rng.standard_normal(d)
    t = time.time()
    A, b, c, K, dAAt, maxb = problems.bqpsos_dual_problem(Q, e, d)
    tg = time.time() - t
    o = {"tol": 1e-8, "dAAt": dAAt, "line_search": 1}
    if eig:
        o["eig"] = eig
    t = time.time()
    _, obj, data = solvers.ManiDSDP_unitdiag(A, b, c, K, dict(o), verbose=False)
    ts = time.time() - t
    print("d=%d n=%d m=%d nnz(A)=%d (generated in %.1f s): GPU solve %.2f s (rtr %.2f s, eig %.2f s), %d outer iterations, %d Hess-vecs, "
          "obj %.8f, eta %.1e, status %d, p %d" % (d, K["s"], b.size, A.nnz, tg, ts, data["rtr_seconds"], data["eig_seconds"],
          data["iters"], data["hessvecs"], obj * maxb, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["fac_size"][-1]),
          flush=True)
