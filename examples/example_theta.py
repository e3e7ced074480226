"""Lovasz theta SDP with ManiSDP_unittrace -- the reference's example/example_theta.m:41-55 on an SDPLIB instance
(theta1 / theta2 are shipped; the optimum is data/sdplib/README:98-105: 23 and 32.879169): argv = [name, default theta1].
With the reference's default options this family reaches its optimum to 6-7 digits and then leaves through the "Slow
progress" exit at eta ~ 1e-5 (oracle and GPU path alike); the script prints eta as it is."""
import sys
import time

import numpy as np

from _common import GOLDEN, eta
from manisdp_matlab_amd import problems, solvers

name = sys.argv[1] if len(sys.argv) > 1 else "theta1"
At, b, c, K = problems.from_sdpa("%s/%s.dat-s.gz" % (GOLDEN, name))
t = time.time()
Y, fval, data = solvers.ManiSDP_unittrace(At, b, c, K, {"tol": 1e-8}, rng=np.random.default_rng(4))
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs" % (fval, eta(data), time.time() - t))
