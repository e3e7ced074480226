"""G81 to KKT 1e-8 with the per-run Lanczos statistics of the device escape on stderr (esc_debug) and the solver's own log."""
import os, sys, time
os.environ["MSDP_ESC_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
for rep in range(2):
    t = time.time()
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40, "tol": 1e-8}, verbose=(rep == 1))
    print("rep %d: %.3f s, rtr %.3f, eig %.3f, iters %d, hessvecs %d, dinf %.2e, verifications %s, bound certs %s" % (
        rep, time.time() - t, data["rtr_seconds"], data["eig_seconds"], data["iters"], data["hessvecs"], data["dinf"],
        data.get("eig_verifications"), data.get("eig_bound_certificates")), flush=True)
