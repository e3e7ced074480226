"""G1 (BASELINE config 0: n = 800, p0 = 2 default options) to KKT 1e-8 with and without the entry-parallel CSR gather of the persistent tCG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G1.txt.gz"))
for ep in (1, 0, 1, 0):
    t = time.perf_counter()
    _, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"device_options": {"persist_ep": ep}}, verbose=False)
    print("persist_ep %d: %.3f s (rtr %.3f, eig %.3f), obj %.8f dinf %.1e, %d AL iterations, %d Hess-vecs" %
          (ep, time.perf_counter() - t, data["rtr_seconds"], data["eig_seconds"], obj, data["dinf"], data["iters"], data["hessvecs"]), flush=True)
