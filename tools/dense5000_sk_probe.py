"""Dense C, n = 5000 (BASELINE config 4's shape), p = 32: Hess-vec time against the number of k slices of the contraction (option dense_sk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
p = 32
h = _lib.Handle.dense_synthetic(n, 0, pcap=p)
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h.set_option("dense_sym", 0)
for sk in (0, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 16, 20, 26, 32):
    try:
        h.set_option("dense_sk", sk)
        h.set_point(Y)
        h.bench_hessvec(50)
        ms = min(h.bench_hessvec(200)[0] for _ in range(3))
        print("n=%d p=%d dense_sk %2d: %.1f us (%.2f TB/s of 8 n^2)" % (n, p, sk, ms * 1e3, 8.0 * n * n / ms / 1e9), flush=True)
    except Exception as e:
        print("dense_sk", sk, "failed:", e)
h.close()
