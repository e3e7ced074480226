"""A/B of option persist_slots on G81 (p = 17..32): three row slots on 216 workgroups against four on 160."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
ps = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 24]
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in ps:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    for rep in range(2):
        for slots in (3, 4):
            h = _lib.Handle.onlyunitdiag(C, pcap=p)
            h.set_option("persist_slots", slots)
            h.set_point(Y)
            h.point_snapshot()
            t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
            best = 1e9
            for _ in range(6):
                h.point_restore()
                t0 = time.perf_counter(); st = h.rtr(opts); best = min(best, time.perf_counter() - t0)
            print("p %2d slots %d: trip %.3f us; trustregions() %.3f ms, %d Hess-vecs -> %.0f Hess-vec/s (cost %.10f)" % (p, slots, t, best * 1e3, st.hessvecs, st.hessvecs / best, st.cost), flush=True)
            h.close()
