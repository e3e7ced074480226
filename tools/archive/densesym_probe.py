"""Symmetric dense contraction (msdp_densesym.hip) against the full one (msdp_dense.hip): Hess-vec time per shape of the
workgroup, agreement with the full kernel, run-to-run bit identity.  argv: n p [p ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = int(sys.argv[1])
modes = [("full", 0, 0, 0, 0), ("sym 8x16", 2, 1, 0, 1), ("sym 8x16 db", 2, 1, 0, 2), ("sym 8x32", 2, 2, 0, 1), ("sym 8x32 db", 2, 2, 0, 2),
         ("sym 16x16", 2, 3, 0, 1), ("sym 16x16 db", 2, 3, 0, 2), ("default", 1, 0, 0, 0)]
for p in [int(x) for x in sys.argv[2:]]:
    h = _lib.Handle.dense_synthetic(n, 0, pcap=p)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h.set_point(Y)
    ref = None
    for name, sym, rt, ln, db in modes:
        h.set_option("dense_sym", sym); h.set_option("dense_sym_rt", rt); h.set_option("dense_sym_len", ln); h.set_option("dense_sym_db", db)
        h.set_point(Y)
        H1 = h.hessvec(U); H2 = h.hessvec(U)
        G = h.rgrad()
        if ref is None:
            ref = (H1, G)
        eh = np.linalg.norm(H1 - ref[0]) / np.linalg.norm(ref[0]); eg = np.linalg.norm(G - ref[1]) / np.linalg.norm(ref[1])
        reps = 200 if n <= 8000 else 60
        h.bench_hessvec(30)
        ms, by, fl = h.bench_hessvec(reps)
        print("n=%d p=%d %-16s: %.1f us (%.2f TB/s of 8n^2, %.1f TF)  |H-Hfull|/|H| %.1e  |G-Gfull|/|G| %.1e  bit-repro %s" %
              (n, p, name, ms * 1e3, by / ms / 1e9, fl / ms / 1e9, eh, eg, np.array_equal(H1, H2)), flush=True)
    h.close()
