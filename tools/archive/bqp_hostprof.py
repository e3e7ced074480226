"""cProfile of the host side of a BQP d = 60 solve (start point 1): where the AL bookkeeping time goes."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
d = 60
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
pr = cProfile.Profile()
pr.enable()
Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False, rng=np.random.default_rng(1))
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14)
print(st.getvalue()[:3500])
print("iters", data["iters"], "rtr", data["rtr_seconds"], "eig", data["eig_seconds"])
