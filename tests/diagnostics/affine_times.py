"""Affine entry points at the BASELINE config shapes: K3 = BQP 2nd-order relaxation (ManiSDP_unitdiag),
K4 = theta-like unit-trace SDP with dense C (ManiSDP_unittrace).  Reports Hess-vec device time (SDDMM + adjoint +
two-matrix MFMA contraction + epilogue) and, for small d, a full solve next to the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from manisdp_matlab_amd import _lib, problems, solvers

def hv_time(kind, At, b, c, n, p, sphere):
    h = _lib.Handle.affine(kind, At, b, c, n, pcap=p)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p))
    Y = Y / np.linalg.norm(Y) if sphere else Y / np.linalg.norm(Y, axis=1, keepdims=True)
    h.set_multipliers(np.zeros(len(b)), 1.0)
    h.set_point(Y)
    for _ in range(2):
        ms, _, _ = h.bench_hessvec(50)
    h.close()
    return ms * 1e3

which = sys.argv[1] if len(sys.argv) > 1 else "bqp"
if which == "bqp":
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rng = np.random.default_rng(2)
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2; e = rng.standard_normal(d)
    t = time.time(); At, b, c, K = problems.bqpmom(d, Q, e); tg = time.time() - t
    c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
    n = K["s"]
    print("BQP d=%d: n=%d m=%d nnz(At)=%d (generated in %.1f s)" % (d, n, len(b), At.nnz, tg), flush=True)
    for p in (8, 16, 32):
        us = hv_time(_lib.KIND_UNITDIAG, At, b, c, n, p, False)
        # B ~ 2*nnz(At)*12 + 8n^2*(write+read of AyU) + 8n^2 (eS) + 3*8np
        B = 2 * At.nnz * 12 + 3 * 8 * n * n + 24 * n * p
        print("  p=%d Hess-vec %.1f us  (%.0f GB/s algorithmic)" % (p, us, B / us / 1e3), flush=True)
    if d <= 40:
        for mode in ("host", "device"):
          t = time.time(); Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {"eig": mode}, verbose=False); tg = time.time() - t
          print("  eig=%s eig_s %.2f" % (mode, data["eig_seconds"]), end="")
          print("  GPU solve: obj %.8f eta %.1e iters %d hessvecs %d time %.2f s (rtr %.2f s)" % (
            obj, max(data["gap"], data["pinf"], data["dinf"]), data["iters"], data["hessvecs"], tg, data["rtr_seconds"]), flush=True)
        if d <= 30:
            from oracle import manisdp_ref as R
            t = time.time(); Yr, objr, dr = R.ManiSDP_unitdiag(At, b, c, K, {}); tc = time.time() - t
            print("  oracle   : obj %.8f eta %.1e iters %d hessvecs %d time %.2f s (rtr %.2f s)" % (
                objr, max(dr["gap"], dr["pinf"], dr["dinf"]), dr["iters"], dr["hessvecs"], tc, dr["rtr_seconds"]), flush=True)
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    At, b, c, K = problems.theta_problem(n, ndraws=10 * n, seed=1)
    print("theta-like: n=%d m=%d nnz(At)=%d" % (n, len(b), At.nnz), flush=True)
    for p in (8, 16, 32):
        us = hv_time(_lib.KIND_UNITTRACE, At, b, c, n, p, True)
        B = 2 * At.nnz * 12 + 3 * 8 * n * n + 24 * n * p
        F = 2 * 2.0 * n * n * p
        print("  p=%d Hess-vec %.1f us  (%.0f GB/s algorithmic, %.1f TFLOP/s fp64)" % (p, us, B / us / 1e3, F / us / 1e6), flush=True)
