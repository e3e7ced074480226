"""The chunked tCG trip (option persist = 0: what every rank of a sharded run and every handle beyond the persistent kernel's
reach executes) on G81 and on toroidal grids.  argv: [p ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
ps = [int(x) for x in sys.argv[1:]] or [32]
cases = [("G81", problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz")))] + \
        [("grid %dx%d" % (r, c), problems.toroidal_grid_maxcut(r, c, seed=3)) for r, c in ((200, 400), (500, 500))]
for name, C in cases:
    n = C.shape[0]
    for p in ps:
        rng = np.random.default_rng(0)
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0)
        h.set_point(Y)
        t = min(h.bench_tcg_trip(128) for _ in range(3)) * 1e3
        ms, by, fl = h.bench_hessvec(100)
        print("%s n=%d p=%d: chunked trip %.2f us, stand-alone Hess-vec %.2f us" % (name, n, p, t, ms * 1e3), flush=True)
        h.close()
