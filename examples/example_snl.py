"""Sensor network localization through the generic ManiSDP -- the workload and options of the reference's
example/Sensor_Network_Localization.m:2-49 (n = 10 sensors, one clique of all 2n variables: a moment matrix of order 231,
m = 16 403): argv = [n sensors, default 10] [seed, default 1]."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f, loc = problems.snl_polynomial(n, seed=seed)
At, b, c, K = problems.snl_mom(f, 2 * n)
c = np.asarray(c.todense()).ravel()
maxc = np.abs(c).max()
opts = {"tol": 1e-4, "sigma0": 1, "sigma_min": 1e1, "theta": 1e-3, "TR_maxiter": 8, "line_search": 0, "alpha": 0.01}   # :40-46
t = time.time()
Y, fval, data = solvers.ManiSDP(At, b, c / maxc, K, opts, verbose=False, rng=np.random.default_rng(0))
X = Y @ Y.T
x = X[1:2 * n + 1, 0] / X[0, 0]                            # first-order moments = the sensor positions when the relaxation is tight
err = np.abs(np.concatenate([loc[0], loc[1]]) - x).max()
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs (n = %d, m = %d), rank %d, largest position error of the first-order moments %.1e"
      % (fval * maxc, eta(data), time.time() - t, K["s"], b.size, Y.shape[1], err))
