import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSDP_TIMING"] = "1"
from manisdp_matlab_amd import _lib, problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
t = time.perf_counter()
_, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
print("total", time.perf_counter() - t, data["rtr_seconds"], data["eig_seconds"], data["hessvecs"], data["iters"])
