"""Round 5: cost of a TR iteration outside its tCG trips in the fused launch: trustregions() with inner caps 1, 2, 4, 8 on G81."""
import os, sys, time
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
h.point_snapshot()
for fused in (1, 0):
    h.set_option("fused_rtr", fused)
    for iters in (40, 200):
        res = []
        for mi in (1, 2, 4, 8):
            opts = _lib.default_opts(maxiter=iters, maxinner=mi, tolgradnorm=1e-12)
            best = 1e9
            for _ in range(5):
                h.point_restore()
                t0 = time.perf_counter(); st = h.rtr(opts); dt = time.perf_counter() - t0
                best = min(best, dt)
            res.append((mi, best * 1e6, st.hessvecs, st.iters))
        # least squares: time = a + iters * b + hessvecs * c
        print("fused %d, %d iterations: " % (fused, iters) + "; ".join("cap %d: %.1f us, %d Hess-vecs, %d iters -> %.2f us per iteration" % (m, t, hv, it, t / it) for m, t, hv, it in res), flush=True)
h.close()
