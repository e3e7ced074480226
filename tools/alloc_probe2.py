"""set_point with a start point the runtime has not seen (a fresh NumPy array per handle, the previous one freed) against a re-used one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSDP_TIMING"] = "1"
import numpy as np
from manisdp_matlab_amd import _lib, problems
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, 40)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
for mode in ("same array", "fresh copy, old one freed first", "fresh copy, old ones kept", "same array", "big temporaries freed between"):
    keep = []
    Yc = Y
    for it in range(3):
        if mode.startswith("fresh copy, old one freed"):
            Yc = None
            Yc = Y.copy()
        elif mode.startswith("fresh copy, old ones kept"):
            Yc = Y.copy(); keep.append(Yc)
        elif mode.startswith("big"):
            tmp = [np.ones(5_000_000) for _ in range(4)]; del tmp
        h = _lib.Handle.onlyunitdiag(C, pcap=56)
        t1 = time.perf_counter(); h.set_point(Yc); t2 = time.perf_counter()
        g = h.get_point()
        h.close()
        print("%s: set_point %.2f ms" % (mode, 1e3 * (t2 - t1)), flush=True)
