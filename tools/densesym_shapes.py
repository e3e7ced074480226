"""Symmetric dense contraction: workgroup shapes of round 6 against round 4's.  argv: n p [p ...]
(shape, dense_sym_rt, dense_sym_res, dense_sym_db)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = int(sys.argv[1])
modes = [("full kernel", 0, 0, 0, 0), ("default", 1, 0, 0, 0), ("8 x 32 rows (r4 default)", 2, 2, 0, 0), ("8 x 32 rows, one buffer", 2, 2, 0, 1),
         ("12 x 32 rows", 2, 4, 0, 0), ("12 x 32 rows, one buffer", 2, 4, 0, 1),          ("16 x 16 rows", 2, 3, 0, 0)]
for p in [int(x) for x in sys.argv[2:]]:
    h = _lib.Handle.dense_synthetic(n, 0, pcap=p)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h.set_point(Y)
    ref = None
    for name, sym, rt, res, db in modes:
        h.set_option("dense_sym", sym); h.set_option("dense_sym_rt", rt); h.set_option("dense_sym_res", res); h.set_option("dense_sym_db", db)
        h.set_point(Y)
        H1 = h.hessvec(U); H2 = h.hessvec(U)
        if ref is None:
            ref = H1
        eh = np.linalg.norm(H1 - ref) / np.linalg.norm(ref)
        h.bench_hessvec(30)
        ms = min(h.bench_hessvec(60)[0] for _ in range(3))
        fl = 2.0 * n * n * p
        print("n=%d p=%d %-44s: %7.1f us  %5.1f TF = %.3f of 78.6  |H-Hfull|/|H| %.1e  bit-repro %s" %
              (n, p, name, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 78.6, eh, np.array_equal(H1, H2)), flush=True)
    h.close()
