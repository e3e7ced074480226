"""Chunked tCG trip (3 kernels) at sizes beyond the persistent kernel: time per trip against the streaming traffic of SURVEY.md 8(d)
(S*U bytes + ~10 vector passes)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from manisdp_matlab_amd import _lib, problems
_lib.load()
for (rows, cols, p) in ((500, 500, 32), (1000, 500, 32), (1000, 1000, 16), (1000, 1000, 32), (500, 500, 64)):
    C = problems.toroidal_grid_maxcut(rows, cols, seed=3)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    path = h.tcg_path()
    ms, by, fl = h.bench_hessvec(50)
    t = h.bench_tcg_trip(64)
    vec = n * p * 8.0
    print("n=%d p=%d path=%d: S*U %.1f us = %.2f TB/s | trip %.1f us; S*U bytes + 10 vector passes = %.0f MB -> %.2f TB/s"
          % (n, p, path, ms * 1e3, by / (ms * 1e-3) / 1e12, t * 1e3, (by + 10 * vec) / 1e6, (by + 10 * vec) / (t * 1e-3) / 1e12), flush=True)
    h.close()
