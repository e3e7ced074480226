"""Dense-C Hess-vec (fp64 MFMA S*U + epilogue) device time and roofline fractions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for p in [int(x) for x in (sys.argv[2:] or ["32"])]:
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n)); C = (G + G.T) / (2 * np.sqrt(n)); del G
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    for _ in range(3):
        ms, by, fl = h.bench_hessvec(200)
    print("n=%d p=%d hessvec %.2f us  %.0f GB/s (%.1f%% of 8 TB/s)  %.1f TFLOP/s fp64 (%.1f%% of 78.6)" % (
        n, p, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80, fl / ms / 1e9, fl / ms / 1e9 / 0.786))
    h.close()
