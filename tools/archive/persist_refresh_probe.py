"""Persistent tCG kernel on G81: drift of tCG's invariant Heta = Hess(eta) and time per trip against the refresh period."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in (8, 16, 32, 40):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    for K in (0, 4, 8, 16, 32):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist_refresh", K); h.set_option("fused_rtr", 0)
        h.set_point(Y)
        h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        Yc = h.get_point()
        out = []
        for trips in (50, 100):
            h.set_point(Yc)
            o = _lib.default_opts(maxiter=1, maxinner=trips, tolgradnorm=1e-14)
            o.Delta0 = 1e3; o.Delta_bar = 1e6
            st = h.rtr(o)
            eta, heta = h.debug_get_tcg_step()
            h.set_point(Yc); h.cost()
            He = h.hessvec(eta)
            out.append((st.hessvecs, np.linalg.norm(heta - He) / np.linalg.norm(heta)))
        h.set_option("fused_rtr", 1)
        h.set_point(Y)
        trip = min(h.bench_tcg_trip(512) for _ in range(3))
        print("p=%d refresh=%2d: trip %.2f us (path %d); drift %s" % (p, K, trip * 1e3, h.tcg_path(), ", ".join("%d trips %.1e" % o for o in out)), flush=True)
        h.close()
