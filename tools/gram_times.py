"""BQP d = 60 Hess-vec at the ranks the solve actually runs at (p up to 300), MFMA vs VALU Gram kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
d = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
n = K["s"]
for p in (32, 96, 200, 300):
    for valu in ("0", "1"):
        os.environ["MSDP_GRAM_VALU"] = valu
        h = _lib.Handle.affine(_lib.KIND_UNITDIAG, At, b, c, n, pcap=p)
        rng = np.random.default_rng(0)
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h.set_multipliers(np.zeros(len(b)), 1.0)
        h.set_point(Y)
        for _ in range(2):
            ms, _, _ = h.bench_hessvec(50)
        h.close()
        print("p=%d gram=%s Hess-vec %.1f us" % (p, "valu" if valu == "1" else "mfma", ms * 1e3), flush=True)
