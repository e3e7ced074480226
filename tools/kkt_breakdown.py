"""Where the G81 solve to KKT 1e-8 (p0 = 40) spends its time: every trustregions() call (width, TR iterations, Hess-vecs, seconds, the tCG
form that ran) and every escape call, second solve of the process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
rec = []
_rtr, _esc = _lib.Handle.rtr, _lib.Handle.escape_eigs
def rtr(self, opts):
    t = time.perf_counter(); st = _rtr(self, opts); dt = time.perf_counter() - t
    rec.append("rtr      p = %2d: %3d TR iterations, %5d Hess-vecs, %7.2f ms wall (%.2f us per Hess-vec), tcg_path %d form %d" %
               (self.p, st.iters, st.hessvecs, 1e3 * dt, 1e6 * dt / max(st.hessvecs, 1), self.tcg_path(), self.persist_form()))
    return st
def esc(self, k, tol=1e-9, maxit=60000):
    t = time.perf_counter(); out = _esc(self, k, tol=tol, maxit=maxit); dt = time.perf_counter() - t
    rec.append("escape   k = %d: %7.2f ms wall, method %d" % (k, 1e3 * dt, self.escape_method()))
    return out
_lib.Handle.rtr, _lib.Handle.escape_eigs = rtr, esc
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
rec.clear()
t = time.perf_counter()
_, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
tot = time.perf_counter() - t
print("\n".join(rec))
print("total %.1f ms: rtr %.1f, escape %.1f, the rest %.1f" % (1e3 * tot, 1e3 * data["rtr_seconds"], 1e3 * data["eig_seconds"], 1e3 * (tot - data["rtr_seconds"] - data["eig_seconds"])))
for row in data["log"]:
    print("AL %d: obj %.8f dinf %.1e r %d p %d t %.3f s hv %d" % row)
