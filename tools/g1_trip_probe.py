"""tCG trip of the persistent kernel on G1 (n = 800, ~48 entries per row: CSR rows, the two-reduction trip) and on G81 for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("G1.txt.gz", "G81.txt.gz"):
    f = os.path.join(root, name)
    if not os.path.exists(f):
        continue
    C = problems.maxcut_cost_matrix(f)
    n = C.shape[0]
    for p in (8, 16, 32):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        Y = np.random.default_rng(0).standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h.set_point(Y)
        t = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
        print("%s n=%d p=%d: tcg_path %d form %d trip %.2f us" % (name, n, p, h.tcg_path(), h.persist_form(), t), flush=True)
        h.close()
