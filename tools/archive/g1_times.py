"""Config 1 (Gset G1, n = 800): tCG trip time and full solve, persistent vs chunked path (MSDP_NO_PERSIST=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G1.txt.gz"))
n = C.shape[0]
for p in (8, 16, 32):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    for _ in range(2): t = h.bench_tcg_trip(512)
    print("G1 p=%d  tCG trip %.2f us  path=%d" % (p, t * 1e3, h.tcg_path()))
    h.close()
t = time.time()
Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {}, verbose=False)
print("G1 solve: obj %.6f dinf %.1e iters %d hessvecs %d  %.3f s (rtr %.3f s, eig %.3f s)" % (
    obj, data["dinf"], data["iters"], data["hessvecs"], time.time() - t, data["rtr_seconds"], data["eig_seconds"]))
