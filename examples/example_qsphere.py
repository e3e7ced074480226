"""Quartic polynomial on the unit sphere, second-order moment relaxation through the generic ManiSDP -- the reference's
example/example_qsphere.m:3-27 (qsmom): argv = [d, default 30] (d = 100 is the n = 5151, m = 8.7 M instance of DESIGN.md)."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
coe = np.random.default_rng(5).standard_normal(problems.get_basis(d, 4).shape[1])
At, b, c, K = problems.qsmom(d, coe)
b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
t = time.time()
Y, fval, data = solvers.ManiSDP(At, b, c, K, {"tol": 1e-8, "theta": 1e-2, "tau1": 0.02}, verbose=False)    # example_qsphere.m:21-25
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs (n = %d, m = %d)" % (fval, eta(data), time.time() - t, K["s"], b.size))
