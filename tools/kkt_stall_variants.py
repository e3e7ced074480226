"""Where does the 20 - 30 ms stall of the second solve's first upload come from?  Variants of what happens between two G81 solves.
    python tools/kkt_stall_variants.py <variant>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSDP_TIMING"] = "1"
import numpy as np
from manisdp_matlab_amd import _lib, problems, solvers
v = sys.argv[1]
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
opts = {"p0": 40}
if v == "maxit1":
    opts["AL_maxiter"] = 1
if v == "hosteig_off":
    pass
r1 = solvers.ManiSDP_onlyunitdiag(C, dict(opts), verbose=False)
if v != "keep":
    r1 = None
if v == "sleep":
    time.sleep(0.5)
if v == "dummy":
    n = C.shape[0]
    Y = np.random.default_rng(1).standard_normal((n, 40))
    h = _lib.Handle.onlyunitdiag(C.tocsr(), pcap=56); h.set_point(Y); h.cost(); h.close()
if v == "y0":
    n = C.shape[0]
    Y = np.random.default_rng(0).standard_normal((n, 40)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    opts["Y0"] = Y
print("---- second solve", file=sys.stderr, flush=True)
t = time.perf_counter()
_, obj, data = solvers.ManiSDP_onlyunitdiag(C, dict(opts), verbose=False)
print(v, "total", time.perf_counter() - t, data["rtr_seconds"], data["eig_seconds"], data["hessvecs"], data["iters"])
