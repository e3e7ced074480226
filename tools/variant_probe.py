"""A/B of library BUILDS (and of the polling back-off) on bench.py's step (G81, p from argv, fused trustregions() call):
python tools/variant_probe.py <lib.so | -> [p] [back-off units, comma separated]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
p = int(sys.argv[2]) if len(sys.argv) > 2 else 32
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
h.set_point(Y); h.point_snapshot()
for bo in ([int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]):
    if bo is not None:
        h.set_option("psync_backoff", bo | (bo << 16))
    for _ in range(5):
        h.point_restore(); h.rtr(opts)
    dev = []
    t0 = time.perf_counter()
    for _ in range(20):
        h.point_restore(); st = h.rtr(opts); dev.append(h.last_rtr_device_ms())
    dt = time.perf_counter() - t0
    print("%s p=%d back-off %s: %d Hess-vecs per call, kernel %.1f us (min %.1f), %.0f Hess-vec/s, cost %.10f" % (os.path.basename(_lib.LIB_PATH), p, bo, st.hessvecs, 1e3 * sum(dev) / len(dev), 1e3 * min(dev), 20 * st.hessvecs / dt, st.cost))
h.close()
