"""Round 5: the one-reduction trip on rows of 7 entries (3-D toroidal grid: six neighbours + the diagonal, stored ELL width 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
from manisdp_matlab_amd import _lib
from oracle import manisdp_ref as R, manopt_rtr

def torus3(a, b, c, seed):
    rng = np.random.default_rng(seed)
    n = a * b * c
    idx = np.arange(n).reshape(a, b, c)
    rows, cols, vals = [], [], []
    for ax in range(3):
        j = np.roll(idx, -1, axis=ax)
        w = rng.choice([-1.0, 1.0], size=n)
        rows += [idx.ravel(), j.ravel()]; cols += [j.ravel(), idx.ravel()]; vals += [w, w]
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    return ((sp.diags(np.asarray(abs(A).sum(axis=1)).ravel()) - A) * 0.25).tocsr()

for dims, p in (((8, 9, 10), 12), ((8, 9, 10), 24), ((20, 25, 30), 16), ((20, 25, 40), 32), ((14, 14, 15), 32)):
    C = torus3(*dims, seed=1)
    n = C.shape[0]
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    out = []
    for pipe, fused in ((0, 1), (0, 0), (1, 0), (1, 1)):
        h.set_option("persist_pipe", pipe)
        h.set_option("fused_rtr", fused)
        h.set_point(Y)
        form = h.persist_form() if h.tcg_path() == 1 else -1
        t = min(h.bench_tcg_trip(512) for _ in range(3)) * 1e3 if form >= 0 else float("nan")
        best = 1e9
        for _ in range(4):
            h.set_point(Y)
            t0 = time.perf_counter(); st = h.rtr(_lib.default_opts(maxiter=30, maxinner=60, tolgradnorm=1e-9)); best = min(best, time.perf_counter() - t0)
        out.append("pipe %d fused %d (form %d): trip %.3f us, %d Hess-vecs, %d iters, cost %.12f, %.0f Hess-vec/s" % (pipe, fused, form, t, st.hessvecs, st.iters, st.cost, st.hessvecs / best))
    line = "torus %s n %d p %d: %s" % (dims, n, p, "; ".join(out))
    if n <= 2000:
        prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 30, 60, 1e-9)
        line += "; oracle: %d Hess-vecs, %d iters, cost %.12f" % (info.hessvecs, info.iters, f_ref)
    print(line, flush=True)
    h.close()
