import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import golden_path
from manisdp_matlab_amd import problems, solvers
At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
rng = np.random.default_rng(3)
Y0 = rng.standard_normal((K["s"], 1)); Y0 /= np.linalg.norm(Y0)
for mode in ("host", "device"):
    Y, obj, d = solvers.ManiSDP_unittrace(At, b, c, K, dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, Y0=Y0, eig=mode), verbose=False)
    print(os.environ.get("MSDP_ADJ_FULL", "0"), mode, "obj %.8f gap %.2e pinf %.2e dinf %.2e status %d iters %d" % (obj, d["gap"], d["pinf"], d["dinf"], d["status"], d["iters"]))
