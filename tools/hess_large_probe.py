"""Stand-alone sparse S*U (k_hess_ell_obl) and cost/gradient at n = 10^6 (toroidal grid), chunk order against the windowed order
(option sweep), and the LDS-staged form of round 5 (option window, --window=0,2 [--winlds=KB]).  argv: [side=1000] [p ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ps = [int(x) for x in sys.argv[2:] if not x.startswith('--')] or [32]
ks = [int(x[4:]) for x in sys.argv if x.startswith('--k=')] or [2]
sweeps = [int(x[8:]) for x in sys.argv if x.startswith('--sweep=')] or [0, 2, 3]
windows = [int(v) for x in sys.argv if x.startswith('--window=') for v in x[9:].split(',')] or [0]
winlds = [int(v) for x in sys.argv if x.startswith('--winlds=') for v in x[9:].split(',')] or [0]
C = problems.toroidal_grid_maxcut(side, side, seed=3)
n = C.shape[0]
for p in ps:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    for sweep, k, window, wl in [(a, b, c, e) for a in sweeps for b in ks for c in windows for e in (winlds if c else [0])]:
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("sweep", sweep)
        h.set_option("window", window)
        if wl:
            h.set_option("window_lds", wl)
        h.set_option("sweep_k", k)
        if "--nograph" in sys.argv:
            h.set_option("graph", 0)
        h.set_point(Y)
        h.bench_hessvec(20)
        ms, by, fl = h.bench_hessvec(100)
        print("n=%d p=%d sweep=%d k=%d window=%d lds=%d: Hess-vec kernel %.1f us, %.2f TB/s algorithmic = %.3f of HBM" % (n, p, sweep, k, window, wl, ms * 1e3, by / ms / 1e9, by / ms / 8e9), flush=True)
        h.close()
