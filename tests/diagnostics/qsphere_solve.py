"""BASELINE config 4's named workload: quartic on the sphere (qsmom, second-order moment relaxation) through the generic
ManiSDP entry point, as example/example_qsphere.m does.  argv: d [oracle]  (random coefficients, seed 5, for d != 10)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from manisdp_matlab_amd import problems as P
d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
coe = np.random.default_rng(5).standard_normal(P.get_basis(d, 4).shape[1])
t = time.time(); At, b, c, K = P.qsmom(d, coe); tg = time.time() - t
b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
print("qsmom d=%d: n=%d m=%d nnz(At)=%d (generated in %.1f s)" % (d, K["s"], len(b), At.nnz, tg), flush=True)
if len(sys.argv) > 2:
    from oracle import manisdp_ref as R
    t = time.time(); Y, obj, data = R.ManiSDP(At, b, c, K, {}, verbose=False); tt = time.time() - t
    print("oracle: obj %.8f eta %.1e status %d iters %d hessvecs %d  %.2f s" % (
        obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], data["hessvecs"], tt), flush=True)
else:
    from manisdp_matlab_amd import solvers
    for mode in ("device",):
        t = time.time(); Y, obj, data = solvers.ManiSDP(At, b, c, K, {"eig": mode}, verbose=False); tt = time.time() - t
        print("GPU (eig=%s): obj %.8f eta %.1e status %d iters %d hessvecs %d  %.2f s (rtr %.2f s, eig %.2f s)" % (
            mode, obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], data["hessvecs"], tt,
            data["rtr_seconds"], data["eig_seconds"]), flush=True)
