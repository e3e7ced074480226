"""G1 (n = 800, default options) to KKT 1e-8 with the saddle escape on the host (the reference's eig(S): the default up to n = 3000) and on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G1.txt.gz"))
for mode, extra in (("host", {}), ("device", {}), ("device", {"device_options": {"escape_method": 2}}), ("host", {}), ("device", {})):
    t = time.perf_counter()
    _, obj, data = solvers.ManiSDP_onlyunitdiag(C, dict({"eig": mode}, **extra), verbose=False)
    print("eig %-6s %s: %.3f s (rtr %.3f, eig %.3f), obj %.8f dinf %.1e, %d AL iterations, %d Hess-vecs, checks %s" %
          (mode, extra.get("device_options", ""), time.perf_counter() - t, data["rtr_seconds"], data["eig_seconds"], obj, data["dinf"], data["iters"], data["hessvecs"],
           data.get("eig_verifications", 0)), flush=True)
