"""Rotation search with outliers (Wahba problem, truncated least squares) through ManiSDP_unittrace -- the reference's
example/example_rotationsearch.m:10-38.  Its data generator and SDP builder (createWahbaProblem, QUASAR_Problem) belong to the
STRIDE / CertifiablyRobustPerception packages, not to the reference tree: problems.wahba_with_outliers / quasar_problem restate
them from the example's parameters and the QUASAR paper.  argv = [N, default 50] [outlier rate, default 0.5]."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
a, b, R_gt, beta, outliers = problems.wahba_with_outliers(N, rate, seed=1)
At, bv, c, K = problems.quasar_problem(a, b, beta ** 2)
t = time.time()
# b / (N + 1): the solver works on X = Z / (N + 1), tr X = 1 (example_rotationsearch.m:37); the sigma schedule is not the default
# one (sigma_min = 1e2 ... 1e7 ends in "Slow progress" on this scaling of the data)
Y, fval, data = solvers.ManiSDP_unittrace(At, bv / (N + 1), c, K, {"tol": 1e-8, "sigma0": 1.0, "sigma_min": 1.0, "sigma_max": 1e4, "eig": "host"},
                                          verbose=False)
X = Y @ Y.T
R, theta = problems.quasar_recover(X, N)
w = np.linalg.eigvalsh(X)
angle = np.degrees(np.arccos(np.clip((np.trace(R.T @ R_gt) - 1.0) / 2.0, -1.0, 1.0)))
tls = sum(min(np.sum((bi - R @ ai) ** 2) / beta ** 2, 1.0) for ai, bi in zip(a, b))
print("QUASAR: N = %d (%d outliers), n = %d, m = %d" % (N, outliers.sum(), K["s"], At.shape[1]))
print("ManiSDP: optimum = %.8f (TLS cost at the recovered rotation %.8f), eta = %.1e, status = %d, time = %.2fs"
      % (fval * (N + 1), tls, eta(data), data["status"], time.time() - t))
print("rank-one gap %.1e, rotation error %.3f deg, inlier set recovered: %s" % (w[-2] / w[-1], angle, np.array_equal(theta > 0, ~outliers)))
