"""Trip time of the cross-rank persistent tCG (msdp_persist.hip XR: one combined launch for N in-process ranks on one GPU) on the
weak-scaled G81 family (toroidal grid, 20 000 rows per rank) against the lock-step chunked trips (option xpersist = 0) and one
unsharded handle.  argv: [N=2] [p=32]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
p = int(sys.argv[2]) if len(sys.argv) > 2 else 32
C = problems.toroidal_grid_maxcut(100 * N, 200, seed=81)
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
grp = [7000]
def run(xp):
    grp[0] += 1
    out = [None] * N
    def body(r):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.comm_init_local(N, r, grp[0])
        h.set_option("xpersist", xp)
        h.set_point(Y)
        c0 = h.collective_calls()
        t = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
        calls = h.collective_calls() - c0
        path = h.tcg_path()
        h.set_point(Y)
        st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        out[r] = (t, path, calls, st.hessvecs, st.seconds)
        h.close()
    th = [threading.Thread(target=body, args=(r,)) for r in range(N)]
    [t.start() for t in th]; [t.join() for t in th]
    return out
for xp in (1, 0):
    res = run(xp)
    print("N=%d n=%d p=%d xpersist=%d: trip %.2f us (path %d), collective calls of 3 x 256 bench trips %d, RTR call: %d Hess-vecs in %.1f ms = %.1f us per Hess-vec" %
          (N, n, p, xp, max(r[0] for r in res), res[0][1], res[0][2], res[0][3], res[0][4] * 1e3, res[0][4] * 1e6 / max(res[0][3], 1)), flush=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_point(Y)
t = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
path = h.tcg_path()
h.set_point(Y)
st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
print("one handle n=%d p=%d: trip %.2f us (path %d), RTR call: %d Hess-vecs in %.1f ms = %.1f us per Hess-vec" % (n, p, t, path, st.hessvecs, st.seconds * 1e3, st.seconds * 1e6 / max(st.hessvecs, 1)), flush=True)
h.close()
