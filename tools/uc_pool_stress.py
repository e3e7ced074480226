"""Root-cause probe for the round-3 corruption of later handles by uncached device memory that was hipFree'd per handle.
Run with MSDP_UC_POOL=0 (one driver block per request, hipFree at destroy: the round-3 arrangement) and without (arena pool);
round 5: 4 = direct + hipMemset + device synchronisation before every hipFree, 5 = direct + every line rewritten with cached stores and
an L2 write-back / invalidate before hipFree, 6 = direct with FINE-GRAINED instead of uncached memory, 7 = arenas of fine-grained memory:
counts the handles whose cost / eG / gradient differ from NumPy.  argv: [iterations=300] [--sync] (hipDeviceSynchronize between handles
via torch-free means: a blocking get_z) """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
rng = np.random.default_rng(5)
grids = [(24, 32), (50, 61), (100, 100), (141, 142), (100, 300), (200, 300)]
sparse = [problems.toroidal_grid_maxcut(r, c, seed=r) for r, c in grids]
bad, alive = 0, []
for it in range(iters):
    C = sparse[int(rng.integers(len(sparse)))]
    n, p = C.shape[0], int(rng.integers(2, 41))
    r2 = np.random.default_rng(it)
    Y = r2.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    CY = C @ Y
    z = np.sum(CY * Y, axis=1)
    e1 = abs(h.cost() - 0.5 * z.sum()) / max(1.0, abs(0.5 * z.sum()))
    e2 = np.linalg.norm(h.get_z() - z) / np.linalg.norm(z)
    e3 = np.linalg.norm(h.rgrad() - (CY - Y * z[:, None])) / np.linalg.norm(CY)
    st = h.rtr(_lib.default_opts(maxiter=3, maxinner=12, tolgradnorm=1e-9))
    ok = e1 < 1e-12 and e2 < 1e-12 and e3 < 1e-12 and np.isfinite(st.cost)
    if not ok:
        bad += 1
        print("iteration %d n=%d p=%d: cost err %.1e eG err %.1e grad err %.1e rtr cost %r" % (it, n, p, e1, e2, e3, st.cost), flush=True)
    alive.append(h)
    while len(alive) > int(rng.integers(1, 5)):
        alive.pop(int(rng.integers(len(alive)))).close()
for h in alive:
    h.close()
print("MSDP_UC_POOL=%s: %d of %d handles wrong; pool stats %s" % (os.environ.get("MSDP_UC_POOL", "1"), bad, iters, _lib.pool_stats()), flush=True)
