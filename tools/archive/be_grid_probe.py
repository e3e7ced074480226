"""Workgroups of the block eigen-solver's filter step at large n (option be_grid): time per filter step of one cold call."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manisdp_matlab_amd import _lib, problems
_lib.load()
for N in (2, 4, 8):
    C = problems.toroidal_grid_maxcut(100 * N, 200, seed=81)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    line = []
    for G in (0, 256, 512):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("escape_deflate", 0); h.set_option("escape_warm", 0); h.set_option("be_grid", G); h.set_option("be_degree", 400)
        h.set_point(Y)
        h.rtr(_lib.default_opts(maxiter=30, maxinner=60, tolgradnorm=1e-6))
        h.escape_eigs(1, tol=1e-9, maxit=3000)
        t0 = time.perf_counter()
        lam, V, lmax, steps = h.escape_eigs(1, tol=1e-9, maxit=3000)
        dt = time.perf_counter() - t0
        line.append("be_grid=%s: %d steps in %.1f ms = %.1f us per step (lam %.6e)" % (G or "auto", steps, dt * 1e3, dt * 1e6 / max(steps, 1), lam[0]))
        h.close()
    print("n=%d  " % n + " | ".join(line), flush=True)
