"""Two-launch against three-launch tCG trip of the chunked path: G81 with the persistent kernel switched off, and toroidal grids
beyond its reach (n = 250 000, 10^6)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import _lib, problems
_lib.load()
g81 = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
cases = [("G81", g81, 8), ("G81", g81, 16), ("G81", g81, 32), ("G81", g81, 64), ("G81", g81, 128)]
for (rows, cols, p) in ((500, 500, 32), (1000, 1000, 16), (1000, 1000, 32), (500, 500, 64)):
    cases.append(("grid %dx%d" % (rows, cols), problems.toroidal_grid_maxcut(rows, cols, seed=3), p))
for name, C, p in cases:
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    res = []
    for trip2 in (1, 0):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0); h.set_option("trip1", 0); h.set_option("trip2", 2 * trip2)
        h.set_point(Y)
        t = min(h.bench_tcg_trip(64) for _ in range(3))
        ms, by, fl = h.bench_hessvec(50)
        res.append(t * 1e3)
        h.close()
    vec = n * p * 8.0
    print("%s n=%d p=%d: two-launch trip %.1f us (12 passes = %.0f MB -> %.2f TB/s), three-launch %.1f us (17 passes -> %.2f TB/s); S*U alone %.1f us"
          % (name, n, p, res[0], 12 * vec / 1e6, 12 * vec / (res[0] * 1e-6) / 1e12, res[1], 17 * vec / (res[1] * 1e-6) / 1e12, ms * 1e3), flush=True)
