"""Per-kernel device times of one tCG trip on the G81-shaped workload (graph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_point(Y)
for _ in range(3):
    h.bench_tcg_trip(2000)   # warm clocks
t = [h.bench_kernel(w, 2000) * 1e3 for w in (0, 1, 2)]
trip = h.bench_tcg_trip(2000) * 1e3
print("p=%d hess %.2f us  upd1 %.2f us  upd2 %.2f us  sum %.2f  trip %.2f us  variant=%s" % (p, t[0], t[1], t[2], sum(t), trip, os.environ.get("MSDP_VARIANT", "")))
