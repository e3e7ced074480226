"""BASELINE config 5 as a WHOLE solve on one MI355X: ManiSDP_onlyunitdiag on the synthetic dense C (n x n generated on the
device, 80 GB at n = 100000; with the fragment-ordered copy 160 GB of the 288 GB), p0 = 64, to KKT 1e-8.
usage: python tools/k5_full_solve.py [n ...]   (default 20000 50000)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from manisdp_matlab_amd import _lib, problems, solvers
_lib.load()
for n in [int(a) for a in sys.argv[1:]] or [20000, 50000]:
    C = problems.SyntheticDenseC(n, seed=0)
    t0 = time.time()
    Y, obj, d = solvers.ManiSDP_onlyunitdiag(C, {"p0": 64, "tol": 1e-8}, verbose=True)
    out = {"n": n, "p_final": int(Y.shape[1]), "obj": obj, "dinf": d["dinf"], "status": d["status"], "AL_iters": d["iters"], "hessvecs": d["hessvecs"],
           "seconds": time.time() - t0, "rtr_seconds": d["rtr_seconds"], "escape_seconds": d["eig_seconds"],
           "independent_lambda_min_checks": d.get("eig_verifications", 0), "matrix_GB": 8e-9 * n * n}
    print(json.dumps(out), flush=True)
    _lib.load().msdp_release_cache()
