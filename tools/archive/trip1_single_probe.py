import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from manisdp_matlab_amd import _lib, problems
_lib.load()
for rows, cols, p in [(500, 500, 32), (1000, 1000, 16), (1000, 1000, 32), (500, 500, 64)]:
    C = problems.toroidal_grid_maxcut(rows, cols, seed=3)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = {}
    for name, opts in (("trip1", {"trip1": 2}), ("trip2", {"trip1": 0, "trip2": 2}), ("three", {"trip1": 0, "trip2": 0})):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0)
        for k, v in opts.items():
            h.set_option(k, v)
        h.set_point(Y)
        out[name] = min(h.bench_tcg_trip(64) for _ in range(3)) * 1e3
        h.close()
    print("grid %dx%d n=%d p=%d:" % (rows, cols, n, p), " ".join("%s %.1f us" % kv for kv in out.items()), flush=True)
