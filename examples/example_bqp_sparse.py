"""Sparse BQP (chain of cliques), sparse second-order moment relaxation through ManiSDP_multiblock -- the reference's
example/example_bqp_sparse.m:3-32 (t = 20 cliques of q = 20 variables: 20 blocks of order 211): argv = [t, default 20] [q, default 20]."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

t = int(sys.argv[1]) if len(sys.argv) > 1 else 20
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cliques, n = problems.chain_cliques(t, q)
coe = np.random.default_rng(1).standard_normal(len(problems.bqp_sparse_monomials(cliques)))
t0 = time.time()
At, b, c, K = problems.bqpmom_sparse(n, cliques, coe)
tgen = time.time() - t0
t0 = time.time()
Y, fval, data = solvers.ManiSDP_multiblock(At, b, c, K, {"tol": 1e-8, "line_search": 1, "tau1": 1}, verbose=False)   # example_bqp_sparse.m:25-31
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs (%d variables, %d blocks of order %d, m = %d; generated in %.1fs)"
      % (fval, eta(data), time.time() - t0, n, t, K["s"][0], b.size, tgen))
