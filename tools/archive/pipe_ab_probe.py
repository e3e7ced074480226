"""Round 5: A/B of the one-reduction trip's switches on G81: pipe_local (rows of the own workgroup from LDS / registers), the back-off
in front of the first poll.  argv: [p list]"""
import os, sys
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
    h.set_point(Y)
    h.set_option("persist_pipe", 0)
    print("G81 p %2d two reductions: trip %.3f us" % (p, min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3), flush=True)
    h.set_option("persist_pipe", 1)
    for local in (0, 1):
        h.set_option("pipe_local", local)
        for first in (0, 15, 19, 23, 27, 31, 35):
            h.set_option("psync_backoff", 19 | (first << 16))
            t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
            print("G81 p %2d one reduction, pipe_local %d, first sleep %2d (0 = default): trip %.3f us" % (p, local, first, t), flush=True)
    h.set_option("psync_backoff", 19)
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    h.set_option("fused_rtr", 0)
    h.set_point(Y)
    st = h.rtr(opts)
    print("G81 p %2d: trustregions() %d Hess-vecs, cost %.12f, stats %s gradnorm %.3e" % (p, st.hessvecs, st.cost, (st.accepted, st.rejected, st.iters, st.last_stop_inner), st.gradnorm), flush=True)
    h.close()
