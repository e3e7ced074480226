"""G81 to KKT 1e-8 (example_maxcut.m's p0 = 40), verbose: rank and Hess-vecs per AL iteration; pipe on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import solvers, problems, _lib
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
for pipe in (1, 0):
    os.environ["MSDP_NO_PERSIST_PIPE"] = str(1 - pipe)
    t0 = time.perf_counter()
    _, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=(pipe == 1))
    print("pipe env %d: %.4f s, obj %.9f dinf %.3e iters %d hessvecs %d rtr %.4f eig %.4f" % (pipe, time.perf_counter() - t0, obj, data["dinf"], data["iters"], data["hessvecs"], data["rtr_seconds"], data["eig_seconds"]), flush=True)
