"""Lovasz theta SDP with ManiSDP_unittrace -- the reference's example/example_theta.m:41-55 with ITS options (tol = 1e-6,
sigma0 = 1e5, sigma_max = 1e8, line search on) on an SDPLIB instance (theta1 / theta2 are shipped; the optimum is
data/sdplib/README:98-105: 23 and 32.879169): argv = [name, default theta1] [seed, default 4].
The outcome depends on the start point, for the oracle as for the GPU path: theta1 converges for
about half of the starts, theta2 leaves through the reference's "Slow progress" exit at eta ~ 3e-4."""
import sys
import time

import numpy as np

from _common import GOLDEN, eta
from manisdp_matlab_amd import problems, solvers

name = sys.argv[1] if len(sys.argv) > 1 else "theta1"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
At, b, c, K = problems.from_sdpa("%s/%s.dat-s.gz" % (GOLDEN, name))
t = time.time()
Y, fval, data = solvers.ManiSDP_unittrace(At, b, c, K, {"tol": 1e-6, "sigma0": 1e5, "sigma_max": 1e8, "line_search": 1},
                                          rng=np.random.default_rng(seed))
print("ManiSDP: optimum = %.8f, eta = %.1e, status = %d, time = %.2fs" % (fval, eta(data), data["status"], time.time() - t))
