import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manisdp_matlab_amd import _lib, problems
_lib.load()
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n, p = C.shape[0], 40
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
Y32 = np.ascontiguousarray(Y[:, :32] / np.linalg.norm(Y[:, :32], axis=1, keepdims=True))
for rep in range(4):
    t0 = time.perf_counter(); h = _lib.Handle.onlyunitdiag(C); t1 = time.perf_counter()
    h.set_point(Y32); t2 = time.perf_counter()
    h.cost(); t3 = time.perf_counter()
    h.set_point(Y32); t4 = time.perf_counter()
    h.close(); t5 = time.perf_counter()
    print("rep %d: create %.1f ms, set_point %.1f ms, cost %.1f ms, second set_point %.1f ms, close %.1f ms" % (rep, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4)), flush=True)
print("--- factor wider than the handle's capacity (p0 = 40 on a handle created for 32), and with head room")
for rep in range(3):
    for pcap in (None, 64):
        t0 = time.perf_counter(); h = _lib.Handle.onlyunitdiag(C) if pcap is None else _lib.Handle.onlyunitdiag(C, pcap=pcap); t1 = time.perf_counter()
        h.set_point(Y); t2 = time.perf_counter()
        h.cost(); t3 = time.perf_counter()
        h.close(); t4 = time.perf_counter()
        print("rep %d pcap %s: create %.1f ms, set_point(p=40) %.1f ms, cost %.1f ms, close %.1f ms" % (rep, pcap, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)), flush=True)
