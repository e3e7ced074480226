"""Number of k slices of the dense contraction (split-K, msdp_dense.hip dense_plan) against the plan's choice: Hess-vec time of
synthetic dense handles.  argv: n p [n p ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
a = [int(x) for x in sys.argv[1:]]
for n, p in zip(a[0::2], a[1::2]):
    h = _lib.Handle.dense_synthetic(n, 0, pcap=p)
    h.set_option("dense_sym", 0)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = []
    for sk in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
        h.set_option("dense_sk", sk)
        h.set_point(Y)
        reps = 200 if n <= 10000 else 60
        h.bench_hessvec(30)
        ms = min(h.bench_hessvec(reps)[0] for _ in range(2))
        out.append("%d:%.1f" % (sk, ms * 1e3))
    print("n=%d p=%d  sk:us  %s" % (n, p, "  ".join(out)), flush=True)
    h.close()
