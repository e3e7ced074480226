"""Trip time of the persistent kernel on Gset graphs with long (CSR) rows: how much of it is synchronisation?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
for name in ("G1.txt.gz", "G11.txt.gz", "G32.txt.gz"):
    f = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.exists(f):
        continue
    C = problems.maxcut_cost_matrix(f)
    n = C.shape[0]
    for p in (16, 32):
        rng = np.random.default_rng(0)
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_point(Y)
        path = h.tcg_path()
        form = h.persist_form() if path == 1 else -1
        t = min(h.bench_tcg_trip(512) for _ in range(3)) * 1e3
        print("%s n %d nnz/row %.1f p %d: path %d form %d trip %.3f us" % (name, n, C.nnz / n, p, path, form, t), flush=True)
        h.close()
