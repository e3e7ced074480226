"""Tiny dense-C Hess-vec run for rocprofv3 --pmc passes (tools/dense_pmc.sh): n = 5000, p from argv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = 5000
for p in [int(x) for x in (sys.argv[1:] or ["64"])]:
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n)); C = (G + G.T) / (2 * np.sqrt(n)); del G
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    h.bench_hessvec(30)
    h.close()
