import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from manisdp_matlab_amd import problems, solvers
gold = "tests/golden"
Q = np.loadtxt(gold + "/bqp_Q_60_1.txt.gz", delimiter=","); e = np.loadtxt(gold + "/bqp_e_60_1.txt.gz", delimiter=",")
At, b, c, K = problems.bqpmom(60, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False)
os.environ["MSDP_TIMING"] = "1"
t = time.time()
Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False)
print("total", time.time() - t, data["rtr_seconds"], data["hessvecs"], file=sys.stderr)
