"""ManiSDP_unittrace with the options of the reference's example/example_theta.m:48-55 (tol 1e-6, sigma0 1e5, sigma_max 1e8,
line search on): SDPLIB theta1 / theta2 and theta-like problems of order n (example_theta.m:2-39 generator).  argv = n [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
opts = {"tol": 1e-6, "sigma0": 1e5, "sigma_max": 1e8, "line_search": 1}
def run(name, At, b, c, K):
    t = time.time()
    Y, fval, data = solvers.ManiSDP_unittrace(At, b, c, K, dict(opts), rng=np.random.default_rng(0), verbose=False)
    print("%s: n=%d m=%d obj %.8f eta %.1e status %d iters %d hessvecs %d  %.1f s (rtr %.1f, eig %.1f)" % (
        name, K["s"], np.asarray(b).size if not hasattr(b, "shape") else b.shape[0], fval, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"],
        data["hessvecs"], time.time() - t, data["rtr_seconds"], data["eig_seconds"]), flush=True)
for name in ("theta1", "theta2"):
    run(name, *problems.from_sdpa(os.path.join(GOLD, name + ".dat-s.gz")))
for n in [int(a) for a in sys.argv[1:]]:
    run("theta-like", *problems.theta_problem(n, ndraws=10 * n, seed=1))
