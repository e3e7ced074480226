"""Round 5: the persistent tCG trip with the neighbours' rows gathered behind per-wave row flags while reduction 2 is in flight
(option persist_early) against the round-4 trip (persist_early = 0).  G81, p in {8, 16, 32}: trip time (bench mode, 512 trips) and whole trustregions() calls (p = 32).
argv: [p list, comma separated]"""
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
    for early in (0, 1, 5, 9, 13, 17, 21, 25, 29, 33):
        h.set_option("persist_early", early)
        t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
        print("p %2d early %2d: trip %.3f us" % (p, early, t), flush=True)
    # psync_backoff interplay with the early trip
    for early in (9, 17):
        h.set_option("persist_early", early)
        for bo in (15, 19, 23):
            h.set_option("psync_backoff", bo)
            t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
            print("p %2d early %d psync_backoff %2d: trip %.3f us" % (p, early, bo, t), flush=True)
    h.set_option("psync_backoff", 19)
    if p == 32:
        opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
        h.point_snapshot()
        for early in (0, 1, 9, 17, 25):
            h.set_option("persist_early", early)
            best, hv, cost = 1e9, 0, 0.0
            for _ in range(6):
                h.point_restore()
                t0 = time.perf_counter(); st = h.rtr(opts); dt = time.perf_counter() - t0
                best = min(best, dt); hv = st.hessvecs; cost = st.cost
            print("p %2d early %2d: trustregions() %.3f ms, %d Hess-vecs -> %.0f Hess-vec/s, cost %.12f, stats %s" %
                  (p, early, best * 1e3, hv, hv / best, cost, (st.accepted, st.rejected, st.iters, st.last_stop_inner)), flush=True)
    h.close()
