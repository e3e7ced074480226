import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import _lib, problems
_lib.load()
g81 = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
g1 = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G1.txt.gz"))
cases = [("G81", g81, p) for p in (8, 16, 32, 64, 128)] + [("G1", g1, p) for p in (8, 40, 80)] + [("grid200x200", problems.toroidal_grid_maxcut(200, 200, seed=1), 40)]
for name, C, p in cases:
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = {}
    for nm, opts in (("linear", {"trip1": 2}), ("two", {"trip1": 0, "trip2": 2}), ("three", {"trip1": 0, "trip2": 0})):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0)
        for k, v in opts.items():
            h.set_option(k, v)
        h.set_point(Y)
        out[nm] = min(h.bench_tcg_trip(64) for _ in range(3)) * 1e3
        h.close()
    print("%s n=%d p=%d:" % (name, n, p), " ".join("%s %.1f us" % kv for kv in out.items()), flush=True)
