"""Persistent tCG kernel against the chunked path beyond n = 32768 (toroidal grids): trip time, Hess-vec counts, cost."""
import os, sys, numpy as np, time
sys.path.insert(0, os.getcwd())
from manisdp_matlab_amd import _lib, problems
_lib.load()
for (rows, cols, p) in ((200, 200, 32), (250, 250, 32), (200, 200, 16), (250, 250, 12), (300, 300, 16), (250, 250, 40)):
    C = problems.toroidal_grid_maxcut(rows, cols, seed=3)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    res = []
    for persist in (1, 0):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", persist)
        h.set_point(Y)
        path = h.tcg_path()
        t = h.bench_tcg_trip(256) * 1e3
        st = h.rtr(_lib.default_opts(maxiter=6, maxinner=40, tolgradnorm=1e-9))
        res.append((path, t, st.hessvecs, st.cost, st.gradnorm))
        h.close()
    print(n, p, "persist path", res[0][0], "trip %.1f us" % res[0][1], "| chunked trip %.1f us" % res[1][1], "| hv", res[0][2], res[1][2], "cost rel", abs(res[0][3]-res[1][3])/abs(res[1][3]))
