"""Workgroup count of the row-parallel launches (option grid) on G81: stand-alone S*U kernel and the chunked tCG trip."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manisdp_matlab_amd import _lib, problems
_lib.load()
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in (8, 16, 32, 64):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    line = []
    for G in (0, 128, 192, 256, 320, 384, 512):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0); h.set_option("grid", G)
        h.set_point(Y)
        ms = min(h.bench_hessvec(200)[0] for _ in range(3))
        trip = min(h.bench_tcg_trip(64) for _ in range(3))
        line.append("G=%s: S*U %.2f us, trip %.1f us" % (G or "auto", ms * 1e3, trip * 1e3))
        h.close()
    print("p=%d  " % p + " | ".join(line), flush=True)
