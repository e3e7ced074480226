"""BQP d = 60: sensitivity of the AL trajectory to last-bits perturbations of the default start point (the reference's
algorithm takes discrete decisions -- rank cuts, sigma doubling/halving, number of escape directions -- on quantities
that differ in the last bits between any two summation orders)."""
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
n = K["s"]
import json
extra = json.loads(os.environ.get("BQP_OPTS", "{}"))            # e.g. BQP_OPTS='{"eig": "host", "dense_eig_max": 100000}'
ks = [int(x) for x in sys.argv[1:]] or list(range(0, 21))
bad = 0
for k in ks:
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, 2))
    if k:
        Y0 += 1e-13 * np.random.default_rng(1000 + k).standard_normal((n, 2))
    Y0 /= np.sqrt(np.sum(Y0 * Y0, axis=1, keepdims=True))
    t = time.time()
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, dict({"Y0": Y0, "AL_maxiter": 160}, **extra), verbose=False)
    eta = max(data["gap"], data["pinf"], data["dinf"])
    bad += data["status"] != 0
    print("gpu pert %d: obj %.8f eta %.1e status %d iters %d hessvecs %d %.1f s" % (
        k, obj, eta, data["status"], data["iters"], data["hessvecs"], time.time() - t), flush=True)
print("not converged within 160 AL iterations: %d of %d" % (bad, len(ks)))
