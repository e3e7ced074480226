"""ManiSDP_multiblock on chains of t cliques of q variables (sparse BQP relaxation): outer iterations and time against t.
usage: python tools/multiblock_scaling.py t q tau1 [t q tau1 ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from manisdp_matlab_amd import problems as P, solvers
a = sys.argv[1:]
for i in range(0, len(a), 3):
    t, q, tau1 = int(a[i]), int(a[i + 1]), float(a[i + 2])
    cl, n = P.chain_cliques(t, q)
    coe = np.random.default_rng(1).standard_normal(len(P.bqp_sparse_monomials(cl)))
    At, b, c, K = P.bqpmom_sparse(n, cl, coe)
    t0 = time.time()
    Y, obj, d = solvers.ManiSDP_multiblock(At, b, c, K, {"tol": 1e-8, "line_search": 1, "tau1": tau1}, verbose=False)
    print(t, q, tau1, "gpu", obj, d["status"], max(d["gap"], d["pinf"], d["dinf"]), d["iters"], "%.2fs rtr %.2f eig %.2f" % (time.time() - t0, d["rtr_seconds"], d["eig_seconds"]), flush=True)
