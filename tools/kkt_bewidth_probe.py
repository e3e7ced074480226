"""G81 to KKT 1e-8 (p0 = 40) with the block width of the escape's eigen-solver (option be_width: 32 / 64 / 128; 0 = planned)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
for w in (0, 32, 128, 0):
    t = time.perf_counter()
    _, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40, "device_options": {"be_width": w}}, verbose=False)
    print("be_width %3d: %.1f ms (rtr %.1f, escape %.1f), %d AL iterations, %d Hess-vecs, obj %.8f, dinf %.2e, p %s, unconverged %s retries %s" %
          (w, 1e3 * (time.perf_counter() - t), 1e3 * data["rtr_seconds"], 1e3 * data["eig_seconds"], data["iters"], data["hessvecs"], obj, data["dinf"],
           [row[4] for row in data["log"]], data.get("eig_unconverged", 0), data.get("eig_retries", 0)), flush=True)
