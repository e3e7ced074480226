"""Which of the first solve's result arrays stalls the next host <-> device copy when it is freed?  A small second handle is the probe:
the latency of its set_point (a 1.3-MB upload + synchronise) after each free."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from manisdp_matlab_amd import _lib, problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
ns = 4000
Cs = sp.random(ns, ns, density=4.0 / ns, random_state=1, format="csr"); Cs = (Cs + Cs.T).tocsr()
hs = _lib.Handle.onlyunitdiag(Cs, pcap=40)
Ys = np.random.default_rng(2).standard_normal((ns, 40)); Ys /= np.linalg.norm(Ys, axis=1, keepdims=True)
def probe(what):
    t = time.perf_counter(); hs.set_point(Ys); hs.cost(); dt = time.perf_counter() - t
    print("%-40s probe %.2f ms" % (what, 1e3 * dt), flush=True)
probe("cold"); probe("warm")
Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
probe("after the solve"); probe("again")
for k in list(data.keys()):
    v = data.pop(k)
    desc = "%s %s" % (type(v).__name__, getattr(v, "shape", ""))
    t = time.perf_counter(); del v; dt = time.perf_counter() - t
    probe("freed data[%r] (%s) in %.2f ms" % (k, desc, 1e3 * dt))
t = time.perf_counter(); del Y; dt = time.perf_counter() - t
probe("freed Y in %.2f ms" % (1e3 * dt))
gc.collect(); probe("gc.collect")
big = np.ones(1_000_000); del big; probe("freed an 8-MB array HIP never saw")
