"""Hess-vec of the two affine BASELINE shapes (BQP d = 60 through ManiSDP_unitdiag, theta n = 5000 through ManiSDP_unittrace) at
p = 32 under the A/B switches of round 4: symmetric contraction (dense_sym) and the B route of A'(A(.)) (affine_broute).
argv: [bqp60] [theta5000] [--profile] (profile: defaults only, graph off, for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
args = sys.argv[1:]
profile = "--profile" in args
nofuse = "--nofuse" in args
quick = "--quick" in args
names = [a for a in args if not a.startswith("--")] or ["bqp60", "theta5000"]
for name in names:
    t0 = time.time()
    if name.startswith("bqp"):
        d = int(name[3:])
        Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=","); e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
        At, b, c, K = problems.bqpmom(d, Q, e); kind = _lib.KIND_UNITDIAG
    else:
        At, b, c, K = problems.theta_problem(int(name[5:]), ndraws=10 * int(name[5:]), seed=1); kind = _lib.KIND_UNITTRACE
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float).ravel()
    n, p = int(K["s"]), 32
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True) if kind == _lib.KIND_UNITDIAG else np.linalg.norm(Y)
    U = rng.standard_normal((n, p))
    h = _lib.Handle.affine(kind, At, b, c, n, pcap=p)
    print("%s: n=%d m=%d nnz=%d  (set-up %.1f s)" % (name, n, b.size, At.nnz, time.time() - t0), flush=True)
    h.set_multipliers(np.zeros(b.size), 1.0)
    ref = None
    combos = [(1, 1, 0, 0)] if profile else ([(1, 1, 0, 0), (1, 1, 0, 1), (1, 0, 0, 0)] if quick else
                                             [(0, 0, 0, 0), (0, 1, 0, 0), (2, 0, 1, 0), (2, 1, 1, 0), (2, 1, 2, 0), (2, 1, 3, 0), (1, 1, 0, 0), (1, 1, 0, 1)])
    sks = [int(a[5:]) for a in args if a.startswith("--sk=")]
    if sks:
        combos = [(1, 1, 0, 0, k) for k in [0] + sks]
    else:
        combos = [c4 + (0,) for c4 in combos]
    for sym, br, rt, ov, sk in combos:
        h.set_option("dense_sk", sk)
        h.set_option("dense_sym", sym); h.set_option("affine_broute", br); h.set_option("dense_sym_rt", rt); h.set_option("affine_overlap", ov)
        if profile:
            h.set_option("graph", 0)
        if nofuse:
            h.set_option("affine_fuse", 0)
        if "--noside" in args:
            h.set_option("affine_side", 0)
        h.set_point(Y)
        H = h.hessvec(U)
        if ref is None:
            ref = H
        err = np.linalg.norm(H - ref) / np.linalg.norm(ref)
        for _ in range(2):
            ms, by, fl = h.bench_hessvec(100)
        print("  dense_sym=%d broute=%d shape=%d overlap=%d sk=%d: %.1f us  (%.2f TB/s algorithmic = %.3f of HBM)  diff vs first %.1e" %
              (sym, br, rt, ov, sk, ms * 1e3, by / ms / 1e9, by / ms / 1e9 / 8.0, err), flush=True)
    h.close()
