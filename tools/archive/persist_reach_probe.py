import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import _lib, problems
_lib.load()
for N, p in [(2, 40), (3, 40), (2, 64), (4, 32), (6, 24), (5, 16), (10, 12)]:
    C = problems.toroidal_grid_maxcut(100 * N, 200, seed=81)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    res = []
    for persist in (1, 0):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", persist)
        h.set_point(Y)
        h.point_snapshot()
        best = None
        for _ in range(3):
            h.point_restore()
            t0 = time.perf_counter()
            st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        trip = h.bench_tcg_trip(64) * 1e3
        res.append((h.tcg_path(), best * 1e3, st.hessvecs, trip, st.cost))
        h.close()
    print("n=%d p=%d: persist path %d RTR %.1f ms (%d Hv) trip %.1f us | chunked path %d RTR %.1f ms (%d Hv) trip %.1f us | cost diff %.1e"
          % (n, p, res[0][0], res[0][1], res[0][2], res[0][3], res[1][0], res[1][1], res[1][2], res[1][3], abs(res[0][4] - res[1][4])), flush=True)
