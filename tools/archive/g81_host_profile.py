"""Where the host-side time of the G81 solve goes (cProfile of the second solve in the process)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
pr.disable()
print("solve %.3f s: rtr %.3f, escape %.3f" % (time.perf_counter() - t, data["rtr_seconds"], data["eig_seconds"]))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
