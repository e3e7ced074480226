"""Quartic on the sphere (second-order moment relaxation, qsmom.m) with d variables through the generic ManiSDP on the GPU, as
example/example_qsphere.m:18-27 sets it up: argv = d [seed].  d = 100 is the n ~ 5000 unit-trace-like workload of BASELINE
config 4 (n = 5151, m = 8.7 M)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems as P, solvers
d = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opts = {"theta": 1e-2, "tau1": 0.02} if "--example-options" in sys.argv else {}      # example_qsphere.m:21-25
t = time.time()
coe = np.random.default_rng(seed).standard_normal(P.get_basis(d, 4).shape[1])
At, b, c, K = P.qsmom(d, coe)
b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
print("d=%d: n=%d m=%d nnz(At)=%d generated in %.1f s" % (d, K["s"], b.size, At.nnz, time.time() - t), flush=True)
t = time.time()
Y, obj, data = solvers.ManiSDP(At, b, c, K, dict(opts), verbose=True)
print("solve %.1f s: obj %.8f eta %.1e status %d iters %d hessvecs %d rtr %.1f s eig %.1f s" % (
    time.time() - t, obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], data["hessvecs"],
    data["rtr_seconds"], data["eig_seconds"]), flush=True)
