"""Second-order moment relaxation of a binary quadratic program with ManiSDP_unitdiag -- the reference's
example/example_bqp.m:3-43 (bqpmom, c scaled by max|c|): argv = [d, default 30] (instances d = 10, 20, 30, 60 are shipped)."""
import sys
import time

import numpy as np

from _common import GOLDEN, eta
from manisdp_matlab_amd import problems, solvers

d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Q = np.loadtxt("%s/bqp_Q_%d_1.txt.gz" % (GOLDEN, d), delimiter=",")
e = np.loadtxt("%s/bqp_e_%d_1.txt.gz" % (GOLDEN, d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel()
mc = np.abs(c).max()
t = time.time()
Y, fval, data = solvers.ManiSDP_unitdiag(At, b, c / mc, K, {"tol": 1e-8})
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs" % (fval * mc, eta(data), time.time() - t))
