"""Per-block storage of the multiblock kind at scale: create time, device memory held, Hess-vec time and a full solve for
direct sums of `nblk` random MaxCut-like blocks of order n (tests/test_gpu_multiblock.py::_stacked_maxcut)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from manisdp_matlab_amd import _lib, solvers


def stacked(nblk, n, seed=0):
    rng = np.random.default_rng(seed)
    C0 = rng.standard_normal((n, n)); C0 = (C0 + C0.T) / 2; np.fill_diagonal(C0, 0.0)
    scale = 1.0 + (np.arange(nblk) % 2)
    c = np.concatenate([(s * C0).reshape(-1) for s in scale])
    At = sp.csc_matrix(([1.0], ([0], [0])), shape=(nblk * n * n, 1))
    return C0, scale, At, np.array([1.0]), c


def main():
    _lib.load()
    for nblk, n in ((1000, 60), (4000, 60), (1000, 200), (16, 2000)):
        C0, scale, At, b, c = stacked(nblk, n)
        N, p = nblk * n, 8
        free0 = _lib.mem_info()[0]
        t0 = time.perf_counter()
        h = _lib.Handle.multiblock(At, b, c, [n] * nblk, nblk)
        tc = time.perf_counter() - t0
        held = free0 - _lib.mem_info()[0]
        rng = np.random.default_rng(1)
        Y = rng.standard_normal((N, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        h.set_multipliers(np.zeros(1), 1.0)
        h.set_point(Y)
        h.cost(); h.rgrad()
        U = h.proj(rng.standard_normal((N, p)))
        h.hessvec(U)
        t0 = time.perf_counter()
        for _ in range(20):
            h.hessvec(U)
        th = (time.perf_counter() - t0) / 20
        h.close()
        t0 = time.perf_counter()
        _, obj, d = solvers.ManiSDP_multiblock(At, b, c, dict(s=[n] * nblk, nob=nblk),
                                               dict(tol=1e-7, p0=[4] * nblk, AL_maxiter=60, seed=0), verbose=False)
        ts = time.perf_counter() - t0
        print(f"{nblk} x {n}: N={N} sum n_i^2={nblk*n*n:.3g} (N^2={N*N:.3g})  create {tc:.3f} s  held {held/2**20:.0f} MiB  "
              f"hessvec(host call, p={p}) {th*1e6:.0f} us  solve {ts:.2f} s status {d['status']} eta "
              f"{max(d['gap'], d['pinf'], d['dinf']):.1e}", flush=True)


main()
