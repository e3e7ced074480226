"""Persistent tCG trip on G81 for a sweep of widths (why is p = 8 slower than p = 16 on the same kernel instance?).  argv: p list"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
ps = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 6, 8, 10, 12, 14, 16, 18, 24, 32]
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in ps:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
    print("p %2d: trip %.3f us" % (p, t), flush=True)
    h.close()
