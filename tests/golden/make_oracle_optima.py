"""Optima the oracle (oracle/manisdp_ref.py, the CPU restatement of the reference's .m files) certifies for the
BASELINE-size instances that are too slow to solve on the CPU inside a test; written to oracle_optima.json and used by
tests/test_gpu_baseline_sizes.py.  The reference stores no optimum for these instances: the
values are pinned by the oracle's own KKT certificate (gap, pinf, dinf < 1e-8 => objective within ~1e-8 relative of
the SDP optimum by weak duality).

Run in the build container (takes ~15 minutes on 4 cores):  python tests/golden/make_oracle_optima.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from manisdp_matlab_amd import problems as P  # noqa: E402
from oracle import manisdp_ref as R  # noqa: E402

GOLD = os.path.dirname(os.path.abspath(__file__))
out = {}

# BQP d = 60, instance 1 of the reference (data/bqp_Q_60_1.txt, bqp_e_60_1.txt; example_bqp.m:36-43), default options,
# default start (NumPy default_rng(0), p0 = 2)
Q = np.loadtxt(os.path.join(GOLD, "bqp_Q_60_1.txt.gz"), delimiter=",")
e = np.loadtxt(os.path.join(GOLD, "bqp_e_60_1.txt.gz"), delimiter=",")
At, b, c, K = P.bqpmom(60, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
n = K["s"]
Y0 = np.random.default_rng(0).standard_normal((n, 2)); Y0 /= np.sqrt(np.sum(Y0 * Y0, axis=1, keepdims=True))
t = time.time()
_, obj, d = R.ManiSDP_unitdiag(At, b, c, K, {"Y0": Y0})
out["bqp60_1"] = {"obj": obj, "eta": max(d["gap"], d["pinf"], d["dinf"]), "status": d["status"], "AL_iters": d["iters"],
                  "hessvecs": d["hessvecs"], "seconds": time.time() - t, "n": n, "m": int(np.asarray(b).size)}
print(out["bqp60_1"], flush=True)

# quartic on the sphere d = 60 (qsmom.m), coefficients default_rng(5) (the reference's qs_c_60_* files are not in its
# tree: .MISSING_LARGE_BLOBS), generic ManiSDP with default options (example_qsphere.m:18-27)
coe = np.random.default_rng(5).standard_normal(P.get_basis(60, 4).shape[1])
At, b, c, K = P.qsmom(60, coe)
b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float)
c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
t = time.time()
_, obj, d = R.ManiSDP(At, b, c, K, {})
out["qsphere60_seed5"] = {"obj": obj, "eta": max(d["gap"], d["pinf"], d["dinf"]), "status": d["status"],
                          "AL_iters": d["iters"], "hessvecs": d["hessvecs"], "seconds": time.time() - t, "n": K["s"],
                          "m": int(b.size)}
print(out["qsphere60_seed5"], flush=True)
json.dump(out, open(os.path.join(GOLD, "oracle_optima.json"), "w"), indent=1)
