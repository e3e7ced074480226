import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
for p0 in (40, 32, 24, 16, 40, 32):
    t = time.time()
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": p0, "tol": 1e-8}, verbose=False)
    print("p0=%d: %.3f s, rtr %.3f, eig %.3f, iters %d, hessvecs %d, dinf %.2e status %d obj %.8f" % (
        p0, time.time() - t, data["rtr_seconds"], data["eig_seconds"], data["iters"], data["hessvecs"], data["dinf"], data["status"], obj), flush=True)
