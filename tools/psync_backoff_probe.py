"""A/B of option psync_backoff on the persistent tCG trip (G81, p = 32): (sleep units before the first poll) | (units after a failed poll) << 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_point(Y)
for first, again in ((0, 0), (4, 0), (8, 0), (12, 0), (16, 0), (8, 1), (8, 2), (12, 2), (16, 4), (0, 2), (0, 4), (24, 4)):
    h.set_option("psync_backoff", first | (again << 8))
    t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
    print("backoff first=%2d again=%d: trip %.3f us" % (first, again, t), flush=True)
h.close()
