"""BQP d = 60: full solves from several start points, MFMA vs VALU Gram kernel (robustness of the AL trajectory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
d = 60
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
for seed in (1, 2, 3, 4):
    for valu in ("0", "1"):
        os.environ["MSDP_GRAM_VALU"] = valu
        t = time.time()
        Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False, rng=np.random.default_rng(seed))
        print("seed %d gram=%s: obj %.8f eta %.1e status %d iters %d  %.1f s" % (
            seed, "valu" if valu == "1" else "mfma", obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], time.time() - t), flush=True)
