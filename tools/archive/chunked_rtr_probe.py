"""Whole trustregions() calls on the chunked path at sizes just beyond the persistent kernel's reach (toroidal 100N x 200 grids):
time per Hess-vec with the linear-product trip (trip1 = 2), the two-launch trip and the three-launch trip.
    python tools/chunked_rtr_probe.py [N p]..."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import _lib, problems
_lib.load()
args = [int(a) for a in sys.argv[1:]]
cases = [tuple(args[i:i + 2]) for i in range(0, len(args), 2)] or [(2, 40), (4, 40), (8, 40)]
for N, p in cases:
    C = problems.toroidal_grid_maxcut(100 * N, 200, seed=81)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    for name, opts in (("linear", {"trip1": 2}), ("two-launch", {"trip1": 0, "trip2": 2}), ("three-launch", {"trip1": 0, "trip2": 0})):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        for k, v in opts.items():
            h.set_option(k, v)
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
        print("n=%d p=%d %-12s path %d: RTR %.1f ms, %d Hess-vecs in %d TR iterations = %.1f us per Hess-vec (trip alone %.1f us), cost %.10f"
              % (n, p, name, h.tcg_path(), best * 1e3, st.hessvecs, st.iters, best * 1e6 / st.hessvecs, trip, st.cost), flush=True)
        h.close()
