"""Rotation search (problems.quasar_problem) at larger N: which AL schedules reach tol.  argv: N [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
for N in [int(x) for x in sys.argv[1:]] or [100, 200]:
    a, b, Rgt, beta, out = problems.wahba_with_outliers(N, 0.5, seed=1)
    At, bv, c, K = problems.quasar_problem(a, b, beta ** 2)
    for opts in (dict(sigma0=1.0, sigma_min=1.0, sigma_max=1e4), dict(tau1=1e-2, tau2=1e-1), dict(sigma0=1.0, sigma_min=1.0, sigma_max=1e4, tau1=1e-2, tau2=1e-1),
                 dict(sigma0=0.1, sigma_min=0.1, sigma_max=1e3), dict(sigma0=1.0, sigma_min=1.0, sigma_max=1e4, AL_maxiter=3000, TR_maxiter=8)):
        t0 = time.time()
        Y, fval, d = solvers.ManiSDP_unittrace(At, bv / (N + 1), c, K, dict(opts, tol=1e-8, eig="host"), verbose=False)
        X = Y @ Y.T
        Rr, th = problems.quasar_recover(X, N)
        tls = sum(min(np.sum((bi - Rr @ ai) ** 2) / beta ** 2, 1.0) for ai, bi in zip(a, b))
        print(N, opts, "status", d["status"], "iters", d.get("iters"), "eta %.1e" % max(d["gap"], d["pinf"], d["dinf"]), "obj %.6f TLS %.6f" % (fval * (N + 1), tls),
              "%.1fs" % (time.time() - t0), flush=True)
