"""A/B of option persist_goff (the byte offsets of the gathers of a persistent tCG trip kept in registers) on G81: trip time and whole
trustregions() calls.  argv: [p list]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
ps = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 16, 8]
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in ps:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    h.point_snapshot()
    for rep in range(2):
        for goff in (0, 1):
            h.set_option("persist_goff", goff)
            t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
            best = 1e9
            for _ in range(6):
                h.point_restore()
                t0 = time.perf_counter(); st = h.rtr(opts); best = min(best, time.perf_counter() - t0)
            print("p %2d goff %d: trip %.3f us; trustregions() %.3f ms, %d Hess-vecs -> %.0f Hess-vec/s" % (p, goff, t, best * 1e3, st.hessvecs, st.hessvecs / best), flush=True)
    h.close()
