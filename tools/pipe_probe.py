"""Round 5: the one-reduction persistent tCG trip (option persist_pipe, msdp_pipe.h) against the two-reduction trip: parity with the
oracle on small grids, the Heta = Hess(eta) invariant on G81, trip time and whole trustregions() calls.  argv: [p list]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
from oracle import manisdp_ref as R, manopt_rtr
ps = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 16, 8]

def rand_point(n, p, seed):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    return Y

# 1. parity on small grids
for shape, p in (((20, 30), 4), ((20, 30), 16), ((33, 37), 20), ((20, 30), 32), ((40, 50), 32)):
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=3)
    n = C.shape[0]
    Y = rand_point(n, p, 11)
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    for pipe in (0, 1):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist_pipe", pipe)
        h.set_point(Y)
        out = []
        for maxinner in (1, 2, 7, 100):
            h.set_point(Y)
            st = h.rtr(_lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
            _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
            out.append((st.hessvecs, info.hessvecs, st.last_stop_inner, info.stop_inner[-1], "%.2e" % abs(st.cost - f_ref), "%.2e" % abs(st.gradnorm - info.gradnorm)))
        h.set_point(Y)
        st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 40, 100, 1e-8)
        print("grid %s p %d pipe %d path %d:" % (shape, p, pipe, h.tcg_path()), out, "full solve: iters %d/%d hv %d/%d cost diff %.2e" %
              (st.iters, info.iters if hasattr(info, "iters") else -1, st.hessvecs, info.hessvecs, abs(st.cost - f_ref)), flush=True)
        h.close()

C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
for p in ps:
    Y = rand_point(n, p, 0)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    for pipe in (0, 1):
        h.set_option("persist_pipe", pipe)
        t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
        print("G81 p %2d pipe %d: trip %.3f us" % (p, pipe, t), flush=True)
    if p >= 16:
        for bo in (0x130013, 0x0b0013, 0x1b0013, 0x130f13):
            h.set_option("psync_backoff", bo)
            t = min(h.bench_tcg_trip(512) for _ in range(4)) * 1e3
            print("G81 p %2d pipe 1 psync_backoff %#x: trip %.3f us" % (p, bo, t), flush=True)
        h.set_option("psync_backoff", 19)
    # 2. invariant Heta = Hess(eta) near a stationary point
    h.set_option("fused_rtr", 0)
    o = _lib.default_opts(maxiter=1, maxinner=100, tolgradnorm=1e-14)
    o.Delta0 = 1e3; o.Delta_bar = 1e6
    h.set_option("persist_pipe", 0)
    h.set_point(Y)
    for _ in range(12):
        h.rtr(_lib.default_opts(maxiter=20, maxinner=100, tolgradnorm=1e-8))
        Yc = h.get_point()
        full = h.rtr(o).hessvecs == 100
        h.set_point(Yc)
        if full:
            break
    for pipe in (0, 1):
        h.set_option("persist_pipe", pipe)
        for trips in (50, 100):
            h.set_point(Yc)
            o.maxinner = trips
            st = h.rtr(o)
            eta, heta = h.debug_get_tcg_step()
            h.set_point(Yc)
            h.cost()
            He = h.hessvec(eta)
            dev = np.linalg.norm(heta - He) / np.linalg.norm(heta)
            print("G81 p %2d pipe %d trips %3d: |Heta - H eta| / |Heta| = %.2e, hv %d stop %d, |eta| %.6e" % (p, pipe, trips, dev, st.hessvecs, st.last_stop_inner, np.linalg.norm(eta)), flush=True)
    # 3. whole solves
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    h.set_option("fused_rtr", 0)
    h.set_point(Y)
    h.point_snapshot()
    for pipe in (0, 1):
        h.set_option("persist_pipe", pipe)
        best, hv, cost = 1e9, 0, 0.0
        for _ in range(5):
            h.point_restore()
            t0 = time.perf_counter(); st = h.rtr(opts); dt = time.perf_counter() - t0
            best = min(best, dt); hv = st.hessvecs; cost = st.cost
        print("G81 p %2d pipe %d: trustregions() %.3f ms, %d Hess-vecs -> %.0f Hess-vec/s, cost %.12f, stats %s gradnorm %.3e" %
              (p, pipe, best * 1e3, hv, hv / best, cost, (st.accepted, st.rejected, st.iters, st.last_stop_inner), st.gradnorm), flush=True)
    h.close()
