"""Soak test of the persistent trips (one reduction where an instance exists, two elsewhere: p = 40, CSR rows): the same trustregions() call many times over, fused and per-iteration launches, p = 8 / 16 / 32,
on G81 and on a small grid (few workgroups): every run must reproduce the first one bit for bit.  argv: [runs = 150]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
cases = [("G81", problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))),
         ("grid 37 x 41", problems.toroidal_grid_maxcut(37, 41, seed=5)), ("grid 141 x 142", problems.toroidal_grid_maxcut(141, 142, seed=6)),
         ("G1", problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G1.txt.gz"))), ("grid 180 x 180", problems.toroidal_grid_maxcut(180, 180, seed=7))]
bad = 0
for name, C in cases:
    n = C.shape[0]
    for p in ((8, 16, 32, 40) if name == "G81" else (8, 16, 32)):
        rng = np.random.default_rng(p)
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        for fused in (1, 0):
            h.set_option("fused_rtr", fused)
            h.set_point(Y)
            if h.tcg_path() != 1:
                print("%s p %d: not persistent" % (name, p)); continue
            form = h.persist_form()
            opts = _lib.default_opts(maxiter=25, maxinner=60, tolgradnorm=1e-9)
            ref = None
            t0 = time.perf_counter()
            for k in range(runs):
                h.set_point(Y)
                st = h.rtr(opts)
                key = (st.cost, st.gradnorm, st.hessvecs, st.iters, st.accepted, st.last_stop_inner)
                if ref is None:
                    ref, Yref = key, h.get_point()
                elif key != ref or (k % 25 == 0 and not np.array_equal(h.get_point(), Yref)):
                    bad += 1
                    print("MISMATCH %s p %d fused %d run %d: %s vs %s" % (name, p, fused, k, key, ref), flush=True)
            print("%s p %d fused %d (trip form %d): %d runs identical (%d Hess-vecs each), %.1f s" % (name, p, fused, form, runs, ref[2], time.perf_counter() - t0), flush=True)
        h.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
