"""G81 from p0 = 2: device lambda_min (deflated / plain Lanczos) against ARPACK shift-invert on the same S, with residuals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from manisdp_matlab_amd import _lib, problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 2, "AL_maxiter": int(sys.argv[1]) if len(sys.argv) > 1 else 12}, verbose=False)
print("solve: obj %.8f dinf %.3e status %d p %d" % (obj, data["dinf"], data["status"], Y.shape[1]))
z = np.asarray(np.sum((C @ Y) * Y, axis=1)).ravel()
S = (C - sp.diags(z)).tocsc()
lam, V = spla.eigsh(S, k=8, sigma=-1e-3, which="LM", tol=1e-13)
o = np.argsort(lam); lam = lam[o]; V = V[:, o]
print("host  lam:", lam, " residuals:", [float(np.linalg.norm(S @ V[:, i] - lam[i] * V[:, i])) for i in range(8)])
G = S @ Y
print("|S Y|_F %.3e  overlap of host v1 with span(Y): %.3e" % (np.linalg.norm(G), np.linalg.norm(np.linalg.qr(Y)[0].T @ V[:, 0])))
h = _lib.Handle.onlyunitdiag(C, pcap=Y.shape[1])
h.set_point(Y)
for (defl, warm, tol, maxit) in [(1, 0, 1e-9, 60000), (0, 0, 1e-9, 60000), (0, 0, 1e-12, 60000), (0, 0, 1e-14, 80000)]:
    h.set_option("escape_deflate", defl); h.set_option("escape_warm", warm)
    t = time.time()
    l, W, lmax, its = h.escape_eigs(4, tol=tol, maxit=maxit)
    nv, conv, res = h.escape_info()
    print("device deflate=%d tol=%.0e: lam %s lmax %.6f steps %d conv %d (%.2f s) true resid %s rayleigh of host v1 %.6e" % (
        defl, tol, l, lmax, its, conv, time.time() - t, [float(np.linalg.norm(S @ W[:, i] - l[i] * W[:, i])) for i in range(nv)],
        float(V[:, 0] @ (S @ V[:, 0]))))
h.close()
