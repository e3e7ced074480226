"""Config 4 end to end: theta-like unit-trace SDP (example/example_theta.m:2-39, options :50-53) of size n on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
At, b, c, K = problems.theta_problem(n, ndraws=10 * n, seed=1)
print("theta-like: n=%d m=%d nnz(At)=%d" % (n, len(b), At.nnz), flush=True)
opts = dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, line_search=1)          # example_theta.m:50-53
if len(sys.argv) > 2: opts["AL_maxiter"] = int(sys.argv[2])
t = time.time()
Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, opts, verbose=False, rng=np.random.default_rng(0))
tt = time.time() - t
print("solve: obj %.8f eta %.1e status %d iters %d hessvecs %d  %.2f s (rtr %.2f s, eig %.2f s, host AL bookkeeping %.2f s)" % (
    obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], data["hessvecs"], tt,
    data["rtr_seconds"], data["eig_seconds"], tt - data["rtr_seconds"] - data["eig_seconds"]), flush=True)
