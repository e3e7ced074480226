"""tCG's invariant Heta = Hess(eta) on G81 for the persistent kernel, the two-launch and the three-launch chunked trips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in (16, 32, 40):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    for name, persist, trip2 in (("persistent", 1, 1), ("two-launch", 0, 1), ("three-launch", 0, 0)):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", persist); h.set_option("fused_rtr", 0); h.set_option("trip2", 2 * trip2)
        h.set_point(Y)
        for warm in (12, 40):
            st0 = h.rtr(_lib.default_opts(maxiter=warm, maxinner=100, tolgradnorm=1e-8))
            Yc = h.get_point()
            for trips in (50, 100):
                h.set_point(Yc)
                o = _lib.default_opts(maxiter=1, maxinner=trips, tolgradnorm=1e-14)
                o.Delta0 = 1e3; o.Delta_bar = 1e6
                st = h.rtr(o)
                eta, heta = h.debug_get_tcg_step()
                h.set_point(Yc); h.cost()
                He = h.hessvec(eta)
                g = h.rgrad()
                print("p=%d %-12s warm=%d gradnorm=%.2e trips=%d: hv=%d stop=%d |Heta-Hess(eta)|/|Heta| = %.2e  (|Heta|=%.2e |grad|=%.2e |eta|=%.2e)" % (
                    p, name, warm, st0.gradnorm, trips, st.hessvecs, st.last_stop_inner, np.linalg.norm(heta - He) / np.linalg.norm(heta),
                    np.linalg.norm(heta), np.linalg.norm(g), np.linalg.norm(eta)), flush=True)
            h.set_point(Yc)
        h.close()
