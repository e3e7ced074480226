"""Per-call cost of the device saddle-escape eigensolver inside a full G81 solve (steps, seconds, us/step)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers, _lib
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
orig = _lib.Handle.escape_eigs
def wrapped(self, k, tol=1e-9, maxit=60000):
    t = time.time()
    out = orig(self, k, tol=tol, maxit=maxit)
    dt = time.time() - t
    print("  escape_eigs k=%d steps=%d  %.3f s  %.1f us/step  lam_min=%.3e" % (k, out[3], dt, dt / max(out[3], 1) * 1e6, out[0][0]), flush=True)
    return out
_lib.Handle.escape_eigs = wrapped
opts = {"p0": int(sys.argv[1]) if len(sys.argv) > 1 else 40}
t = time.time()
Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, opts, verbose=True)
print(json.dumps({"obj": obj, "dinf": data["dinf"], "time": time.time() - t, "rtr_s": data["rtr_seconds"], "eig_s": data["eig_seconds"]}))
