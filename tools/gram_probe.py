"""BQP d = 60 Hess-vecs at one rank p (argv[1], default 300) for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = 60
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
n = K["s"]
h = _lib.Handle.affine(_lib.KIND_UNITDIAG, At, b, c, n, pcap=p)
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h.set_multipliers(np.zeros(len(b)), 1.0)
h.set_point(Y)
ms, _, _ = h.bench_hessvec(50)
print("p=%d Hess-vec %.1f us" % (p, ms * 1e3))
h.close()
