"""BQP d = 60 default-start solve: where the wall-clock goes (RTR, escape, the rest of the host loop), with cProfile of the host side."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
d = 60
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
for rep in range(2):
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False)
    pr.disable()
    print("rep %d: %.2f s total, rtr %.2f, eig %.2f, iters %d, hessvecs %d, status %d" % (rep, time.time() - t, data["rtr_seconds"], data["eig_seconds"], data["iters"], data["hessvecs"], data["status"]), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:5000])
