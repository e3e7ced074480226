"""Config 3 end to end: BQP second-order relaxation (ManiSDP_unitdiag), random instance of size d, full solve on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
d = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
if os.path.exists(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d)):
    Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")       # example_bqp.m:5-6
    e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
else:
    rng = np.random.default_rng(2)
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2; e = rng.standard_normal(d)
t = time.time(); At, b, c, K = problems.bqpmom(d, Q, e); tg = time.time() - t
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
print("BQP d=%d: n=%d m=%d nnz(At)=%d (generated in %.1f s)" % (d, K["s"], len(b), At.nnz, tg), flush=True)
t = time.time()
Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=bool(os.environ.get("BQP_VERBOSE")))
tt = time.time() - t
print("solve: obj %.8f eta %.1e status %d iters %d hessvecs %d  %.2f s (rtr %.2f s, eig %.2f s, host AL bookkeeping %.2f s)" % (
    obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], data["hessvecs"], tt,
    data["rtr_seconds"], data["eig_seconds"], tt - data["rtr_seconds"] - data["eig_seconds"]), flush=True)
