"""Block eigen-solver on a toroidal grid at a random point and after RTR, with the per-round statistics on stderr."""
import os, sys
os.environ["MSDP_ESC_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from manisdp_matlab_amd import _lib, problems
rows, cols, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
C = problems.toroidal_grid_maxcut(rows, cols, seed=9)
n = C.shape[0]
rng = np.random.default_rng(p)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=64)
h.set_point(Y)
for stage in ("random", "rtr"):
    if stage == "rtr":
        h.rtr(_lib.default_opts(maxiter=60, maxinner=200, tolgradnorm=1e-9))
    z = h.get_z()
    w = np.linalg.eigvalsh((C - sp.diags(z)).toarray()) if n <= 8000 else None
    lam, V, lmax, deg = h.escape_eigs(8, tol=1e-9, maxit=60000)
    print(stage, "deg", deg, "info", h.escape_info(), "lam", lam, "lmax", lmax, flush=True)
    if w is not None:
        print("   lapack", w[:8], w[-1], " spectrum index 64/128:", w[64], w[128], flush=True)
