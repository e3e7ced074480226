"""Round 5: whole trustregions() calls on G81 -- fused (one launch) / per-iteration launches x one- / two-reduction trip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
ps = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 16, 8]
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in ps:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    h.set_point(Y)
    h.point_snapshot()
    for fused in (1, 0):
        for pipe in (0, 1):
            h.set_option("fused_rtr", fused)
            h.set_option("persist_pipe", pipe)
            best = 1e9
            for _ in range(6):
                h.point_restore()
                t0 = time.perf_counter(); st = h.rtr(opts); dt = time.perf_counter() - t0
                best = min(best, dt)
            print("G81 p %2d fused %d pipe %d: trustregions() %.3f ms, %d Hess-vecs -> %.0f Hess-vec/s, cost %.12f, stats %s gradnorm %.6e" %
                  (p, fused, pipe, best * 1e3, st.hessvecs, st.hessvecs / best, st.cost, (st.accepted, st.rejected, st.iters, st.last_stop_inner), st.gradnorm), flush=True)
    h.close()
