"""The same BQP through the dual approach: SOS relaxation with ManiDSDP_unitdiag, next to the moment relaxation with
ManiSDP_unitdiag -- the reference's example/dual/example_bqp_dual.m:1-37 (random Q, e; line search on): argv = [d, default 30]."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(1)
Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
e = rng.standard_normal(d)
At, b, c, K = problems.bqpmom(d, Q, e)
t = time.time()
_, fval, data = solvers.ManiSDP_unitdiag(At, b, c, K, {"tol": 1e-8}, verbose=False)
print("ManiSDP : optimum = %.8f, eta = %.1e, time = %.2fs" % (fval, eta(data), time.time() - t))
A, bs, cs, Ks, dAAt, maxb = problems.bqpsos_dual_problem(Q, e, d)
t = time.time()
_, dfval, ddata = solvers.ManiDSDP_unitdiag(A, bs, cs, Ks, {"tol": 1e-8, "dAAt": dAAt, "line_search": 1}, verbose=False)
print("ManiDSDP: optimum = %.8f, eta = %.1e, time = %.2fs" % (dfval * maxb, eta(ddata), time.time() - t))
