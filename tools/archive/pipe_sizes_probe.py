"""Round 5: one-reduction against two-reduction trip over the instances (row slots per workgroup) -- toroidal grids of several sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
for side, p in ((100, 200, 32), (180, 180, 32), (180, 180, 24), (200, 200, 16), (250, 250, 8), (141, 142, 12), (60, 70, 32)) if False else ((( 100, 200), 32), ((180, 180), 32), ((180, 180), 24), ((200, 200), 16), ((250, 250), 8), ((141, 142), 12), ((60, 70), 32)):
    C = problems.toroidal_grid_maxcut(side[0], side[1], seed=3)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    out = []
    for pipe, local in ((0, 0), (1, 0), (1, 1)):
        h.set_option("persist_pipe", pipe)
        h.set_option("pipe_local", local)
        form = h.persist_form() if h.tcg_path() == 1 else -1
        t = min(h.bench_tcg_trip(512) for _ in range(3)) * 1e3
        out.append("pipe %d local %d (form %d): %.3f us" % (pipe, local, form, t))
    print("grid %s n %d p %d: %s" % (side, n, p, "; ".join(out)), flush=True)
    h.close()
