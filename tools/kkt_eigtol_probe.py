"""G81 to KKT 1e-8 (p0 = 40) with the tolerance of the escape's eigen-solver loosened (option eig_tol: regular calls AND the final check in this
probe): how much of the 53-ms escape is the last digits of lambda_min, and does the AL trajectory move?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
for tol in (1e-9, 1e-8, 1e-7, 1e-6, 1e-9):
    t = time.perf_counter()
    _, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40, "eig_tol": tol}, verbose=False)
    print("eig_tol %.0e: %.1f ms (rtr %.1f, escape %.1f), %d AL iterations, %d Hess-vecs, obj %.8f, dinf %.2e, p per iteration %s" %
          (tol, 1e3 * (time.perf_counter() - t), 1e3 * data["rtr_seconds"], 1e3 * data["eig_seconds"], data["iters"], data["hessvecs"], obj, data["dinf"],
           [row[4] for row in data["log"]]), flush=True)
