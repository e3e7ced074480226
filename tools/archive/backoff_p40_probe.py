"""Back-off in front of the first poll of the two-reduction trip at p = 40 (G81): low byte = reduction 1, bits 16..23 = reduction 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in (40, 64):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    for lo in (11, 15, 19, 23, 27):
        for hi in (0, 15, 23, 31):
            h.set_option("psync_backoff", lo | (hi << 16))
            t = min(h.bench_tcg_trip(512) for _ in range(3)) * 1e3
            print("p %d: first sleep reduction 1 = %2d, reduction 2 = %2d (0: the same): trip %.3f us" % (p, lo, hi, t), flush=True)
    h.close()
