"""Gram route: MFMA vs VALU kernel, Hess-vec / gradient / cost agreement at BQP shapes (no oracle needed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
d = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
n = K["s"]
os.environ["MSDP_AFFINE_ROUTE"] = "gram"
for p in [int(x) for x in (sys.argv[2:] or ["40", "130", "200", "257", "300"])]:
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    y = rng.standard_normal(len(b)) * 0.1
    res = {}
    for valu in ("0", "1"):
        os.environ["MSDP_GRAM_VALU"] = valu
        h = _lib.Handle.affine(_lib.KIND_UNITDIAG, At, b, c, n, pcap=p)
        h.set_multipliers(y, 0.7)
        h.set_point(Y)
        res[valu] = (h.cost(), h.rgrad(), h.hessvec(U))
        h.close()
    f0, g0, h0 = res["0"]; f1, g1, h1 = res["1"]
    print("n=%d p=%d: cost diff %.2e  grad relerr %.2e  hess relerr %.2e" % (
        n, p, abs(f0 - f1) / abs(f1), np.linalg.norm(g0 - g1) / np.linalg.norm(g1), np.linalg.norm(h0 - h1) / np.linalg.norm(h1)), flush=True)
