"""Block eigen-solver on the synthetic dense C (config-5 family) at n = argv[1] (default 20000): per-round statistics, bounded budget."""
import os, sys, time
os.environ["MSDP_ESC_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
p = 64
h = _lib.Handle.dense_synthetic(n, 0, pcap=96)
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h.set_point(Y)
for stage in range(2):
    t = time.perf_counter()
    st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
    print("rtr %.2f s, %d Hess-vecs, gradnorm %.2e" % (time.perf_counter() - t, st.hessvecs, st.gradnorm), flush=True)
    for method in (2, 1):
        h.set_option("escape_method", method)
        t = time.perf_counter()
        lam, V, lmax, steps = h.escape_eigs(8, tol=1e-9, maxit=4000)
        print("method %d: %.2f s, %d steps, conv %s, lam0 %.6e lam7 %.6e lmax %.6f" % (method, time.perf_counter() - t, steps, h.escape_info()[1], lam[0], lam[7], lmax), flush=True)
h.close()
