"""Windowed against chunked row traversal of the gather kernels (option sweep) on toroidal grids beyond the persistent kernel's
reach: two-launch tCG trip, stand-alone S*U, block eigen-solver filter step.
    python tools/sweep_probe.py [rows cols p]..."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import _lib, problems
_lib.load()
args = [int(a) for a in sys.argv[1:]]
cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(500, 500, 32), (1000, 1000, 16), (1000, 1000, 32), (500, 500, 64)]
for rows, cols, p in cases:
    C = problems.toroidal_grid_maxcut(rows, cols, seed=3)
    n = C.shape[0]
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = {}
    for sweep in (3, 2, 0):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0); h.set_option("sweep", sweep)
        h.set_point(Y)
        t = min(h.bench_tcg_trip(64) for _ in range(3))
        ms, by, fl = h.bench_hessvec(50)
        out[sweep] = (t * 1e3, ms * 1e3)
        h.close()
    print("grid %dx%d n=%d p=%d: trip windowed+nt %.1f us / windowed %.1f us / chunked %.1f us; S*U %.1f / %.1f / %.1f us"
          % (rows, cols, n, p, out[3][0], out[2][0], out[0][0], out[3][1], out[2][1], out[0][1]), flush=True)
