"""Sparse quartic whose clique sub-vectors lie on unit spheres, sparse second-order moment relaxation through ManiSDP_multiblock
with K.nob = 0 -- the reference's example/example_qsphere_sparse.m:3-32 (t = 10 cliques of q = 10 variables: 10 blocks of
order 66): argv = [t, default 10] [q, default 10]."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

t = int(sys.argv[1]) if len(sys.argv) > 1 else 10
q = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cliques, n = problems.chain_cliques(t, q)
coe = np.random.default_rng(1).standard_normal(len(problems.quartic_sparse_monomials(cliques)))
At, b, c, K = problems.qsmom_sparse(n, cliques, coe)
opts = {"tol": 1e-4, "theta": 1e-4, "tau1": 1e-3, "tau2": 1e-2, "line_search": 0, "alpha": 0.01}     # example_qsphere_sparse.m:25-31
t0 = time.time()
Y, fval, data = solvers.ManiSDP_multiblock(At, b, c, K, opts, verbose=False)
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs (%d variables, %d blocks of order %d, m = %d)"
      % (fval, eta(data), time.time() - t0, n, t, K["s"][0], b.size))
