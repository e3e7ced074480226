"""theta-like unit-trace Hess-vecs (n = 5000, p from argv) for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 5000
At, b, c, K = problems.theta_problem(n, ndraws=10 * n, seed=1)
c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
h = _lib.Handle.affine(_lib.KIND_UNITTRACE, At, b, c, n, pcap=p)
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y)
h.set_multipliers(np.zeros(len(b)), 1.0)
h.set_point(Y)
ms, _, _ = h.bench_hessvec(50)
print("theta n=%d p=%d Hess-vec %.1f us" % (n, p, ms * 1e3))
h.close()
