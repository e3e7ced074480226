"""Round 5: the Heta = Hess(eta) invariant and the trip time of the one-reduction trip against the refresh interval.  argv: [p]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_option("fused_rtr", 0)
o = _lib.default_opts(maxiter=1, maxinner=100, tolgradnorm=1e-14)
o.Delta0 = 1e3; o.Delta_bar = 1e6
h.set_point(Y)
for _ in range(12):
    h.rtr(_lib.default_opts(maxiter=20, maxinner=100, tolgradnorm=1e-8))
    Yc = h.get_point()
    full = h.rtr(o).hessvecs == 100
    h.set_point(Yc)
    if full:
        break
for pipe, refresh in ((0, 32), (1, 32), (1, 16), (1, 8), (1, 4), (1, 0)):
    h.set_option("persist_pipe", pipe)
    h.set_option("persist_refresh", refresh if not pipe else 32)
    h.set_option("pipe_refresh", refresh)
    t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
    out = []
    for trips in (50, 100, 200):
        h.set_point(Yc)
        o.maxinner = trips
        st = h.rtr(o)
        eta, heta = h.debug_get_tcg_step()
        h.set_point(Yc)
        h.cost()
        He = h.hessvec(eta)
        out.append("%d: %.2e (hv %d)" % (trips, np.linalg.norm(heta - He) / np.linalg.norm(heta), st.hessvecs))
    print("p %d pipe %d refresh %2d: trip %.3f us, |Heta - H eta| / |Heta| at trips %s" % (p, pipe, refresh, t, ", ".join(out)), flush=True)
h.close()
