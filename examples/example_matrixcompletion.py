"""Nuclear-norm matrix completion through the generic ManiSDP -- the reference's example/example_matrixcompletion.m:8-61
(p = q = 2000, rank 10, 400 n draws: n = 4000, m = 1.4 M): argv = [p (= q), default 500] [rank, default 10]."""
import sys
import time

import numpy as np

from _common import eta
from manisdp_matlab_amd import problems, solvers

p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
At, b, c, K, M, _ = problems.matrix_completion(p, p, k, seed=3)
t = time.time()
opts = {"tol": 1e-8, "theta": 1e-2, "TR_maxinner": 6, "TR_maxiter": 8, "delta": 10, "alpha": 0.1}   # example_matrixcompletion.m:51-57
Y, fval, data = solvers.ManiSDP(At, b, c, K, opts, verbose=False)
X12 = Y[:p] @ Y[p:].T
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs (n = %d, m = %d), |X12 - M|/|M| = %.1e, rank %d"
      % (fval, eta(data), time.time() - t, K["s"], b.size, np.linalg.norm(X12 - M) / np.linalg.norm(M), Y.shape[1]))
