import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G1.txt.gz"))
n = C.shape[0]
for (p, q) in [(3, 2), (4, 2), (3, 3), (5, 2), (12, 8), (1, 2)]:
    rng = np.random.default_rng(p)
    Y = np.hstack([rng.standard_normal((n, p)), np.zeros((n, q))]); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = np.hstack([np.zeros((n, p)), np.linalg.qr(rng.standard_normal((n, q)))[0]])
    co = lambda Z: float(np.sum((C @ Z) * Z))
    h = _lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    out = []
    for alpha in (1.0, 0.8, 0.2):
        Z = Y + alpha * U; Z /= np.linalg.norm(Z, axis=1, keepdims=True)
        out.append(abs(h.linesearch_cost(U, alpha) - co(Z)) / abs(co(Z)))
    R = h.retr(U)
    Z = Y + U; Z /= np.linalg.norm(Z, axis=1, keepdims=True)
    print(p, q, ["%.1e" % v for v in out], "retr err %.1e" % (np.linalg.norm(R - Z) / np.linalg.norm(Z)), "cost(Y) err %.1e" % (abs(h.linesearch_cost(None, 0.0) - co(Y)) / abs(co(Y))))
    h.close()
