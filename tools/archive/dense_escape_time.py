"""Device escape eigensolver on a dense S (n = 5000 by default): steps, seconds, us per Lanczos step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
p = 32
rng = np.random.default_rng(0)
G = rng.standard_normal((n, n)); C = (G + G.T) / (2 * np.sqrt(n)); del G
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_point(Y)
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-9
maxit = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
for k in (1, 8):
    for rep in range(2):
        t = time.time()
        out = h.escape_eigs(k, tol=tol, maxit=maxit)
        dt = time.time() - t
        print("k=%d steps=%d %.3f s  %.1f us/step lam_min=%.6f" % (k, out[3], dt, dt / max(out[3], 1) * 1e6, out[0][0]), flush=True)
h.close()
