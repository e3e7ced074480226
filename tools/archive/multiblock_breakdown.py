"""Sparse BQP through ManiSDP_multiblock (example_bqp_sparse.m's size by default): where the wall-clock goes."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
t = int(sys.argv[1]) if len(sys.argv) > 1 else 20
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cl, n = problems.chain_cliques(t, q)
coe = np.random.default_rng(1).standard_normal(len(problems.bqp_sparse_monomials(cl)))
At, b, c, K = problems.bqpmom_sparse(n, cl, coe)
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    Y, obj, d = solvers.ManiSDP_multiblock(At, b, c, K, {"tol": 1e-8, "line_search": 1, "tau1": 1}, verbose=False)
    pr.disable()
    print("rep %d: %.2f s, rtr %.2f, eig %.2f, iters %d, hessvecs %d, cost evals %d, status %d" % (rep, time.time() - t0, d["rtr_seconds"], d["eig_seconds"], d["iters"], d["hessvecs"], d["cost_evals"], d["status"]), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue()[:4000])
