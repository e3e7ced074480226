"""G81 KKT solve: per trustregions() call -- p, TR iterations, Hess-vecs, seconds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import solvers, problems, _lib
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
log = []
_orig = _lib.Handle.rtr
def _rtr(self, opts):
    t0 = time.perf_counter(); st = _orig(self, opts); log.append((self.p, st.iters, st.hessvecs, st.accepted, st.rejected, (time.perf_counter() - t0) * 1e3, st.seconds * 1e3, self.tcg_path(), self.persist_form()))
    return st
_lib.Handle.rtr = _rtr
t0 = time.perf_counter(); _, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False); dt = time.perf_counter() - t0
print("total %.1f ms, rtr %.1f, eig %.1f" % (dt * 1e3, data["rtr_seconds"] * 1e3, data["eig_seconds"] * 1e3))
for r in log:
    print("p %d: %d TR iterations, %d Hess-vecs (%d accepted, %d rejected), %.2f ms wall (%.2f ms reported), path %d form %d -> %.2f us per Hess-vec, %.1f Hess-vecs per iteration" % (r + (r[5] * 1e3 / max(r[2], 1), r[2] / max(r[1], 1))))
