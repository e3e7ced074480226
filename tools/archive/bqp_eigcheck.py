"""BQP d = 60, default start: compare the device escape (lambda_min, lambda_max, escape directions) with LAPACK on the
same dual slack S at every AL iteration of a window (diagnostic for the stall of VERDICT round 1, weak item 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers

d = int(os.environ.get("BQP_D", "60"))
lo, hi = int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 90
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()

def hook(L):
    it = L["it"]
    if it < lo or it > hi:
        return
    h = L["h"]
    S = h.get_dual_slack()
    w, V = np.linalg.eigh(S)
    lam, vS = L["lam"], L["vS"]
    nv, conv, res = h.escape_info()
    # angle between the device escape subspace and the true bottom eigenspace of the same dimension
    k = int(np.sum(np.isfinite(lam)))
    P = V[:, :k].T @ vS[:, :k]
    sv = np.linalg.svd(P, compute_uv=False)
    resid = [np.linalg.norm(S @ vS[:, i] - lam[i] * vS[:, i]) for i in range(k)]
    print("  [eigcheck it %d] host lam[:8] %s\n                  dev  lam[:8] %s  lmax host %.6f dev %.6f  nvalid %d conv %d  min cos %.3e  max resid %.2e  sym %.1e"
          % (it, np.array2string(w[:8], precision=6), np.array2string(lam, precision=6), w[-1], L["lam_max"], nv, conv,
             sv.min(), max(resid), np.abs(S - S.T).max()), flush=True)

Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {"iter_hook": hook, "AL_maxiter": hi + 1}, verbose=True)
