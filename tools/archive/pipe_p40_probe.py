"""Round 5: G81 at p = 40 / 48 / 64 (32 lanes per row): one- against two-reduction trip, whole trustregions() calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
ps = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [40, 64]
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in ps:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    h.set_point(Y)
    h.point_snapshot()
    for pipe in (0, 1):
        h.set_option("persist_pipe", pipe)
        t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
        best = 1e9
        for _ in range(5):
            h.point_restore()
            t0 = time.perf_counter(); st = h.rtr(opts); dt = time.perf_counter() - t0
            best = min(best, dt)
        print("G81 p %2d pipe %d (form %d): trip %.3f us; trustregions() %.3f ms, %d Hess-vecs -> %.0f Hess-vec/s, cost %.12f, stats %s gradnorm %.6e" %
              (p, pipe, h.persist_form(), t, best * 1e3, st.hessvecs, st.hessvecs / best, st.cost, (st.accepted, st.rejected, st.iters, st.last_stop_inner), st.gradnorm), flush=True)
    h.close()
