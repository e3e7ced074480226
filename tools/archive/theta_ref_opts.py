import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from manisdp_matlab_amd import problems, solvers
GOLD = os.path.join(os.getcwd(), "tests", "golden")
opts = {"tol": 1e-6, "sigma0": 1e5, "sigma_max": 1e8, "line_search": 1}
for name in ("theta1", "theta2"):
    At, b, c, K = problems.from_sdpa(os.path.join(GOLD, name + ".dat-s.gz"))
    for seed in range(6):
        for eig in ("host", "device"):
            Y, fval, d = solvers.ManiSDP_unittrace(At, b, c, K, dict(opts, eig=eig), rng=np.random.default_rng(seed), verbose=False)
            print(name, seed, eig, "obj %.8f eta %.1e status %d iters %d" % (fval, max(d["gap"], d["pinf"], d["dinf"]), d["status"], d["iters"]), flush=True)
