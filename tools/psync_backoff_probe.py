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
grid = [(f1, 0) for f1 in (0, 12, 14, 16, 18, 19, 20, 22, 24, 26)] + [(f1, f3) for f1 in (16, 19, 22) for f3 in (14, 17, 22, 25)]
for f1, f3 in grid:                        # f1: all reductions (the one-value ones when f3 is set); f3: the three-value reductions' own figure
    h.set_option("psync_backoff", f1 | (f3 << 16))
    t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
    print("backoff %2d / three-value %2d: trip %.3f us" % (f1, f3, t), flush=True)
h.close()
