"""theta1 / theta2 (SDPLIB, options of example_theta.m:50-53) from several start points: GPU path next to the oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
from oracle import manisdp_ref as R
gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
known = json.load(open(os.path.join(gold, "known_answers.json")))
for case in sys.argv[1:] or ["theta1", "theta2"]:
    At, b, c, K = problems.from_sdpa(os.path.join(gold, case + ".dat-s.gz"))
    c = np.asarray(c.todense()).ravel(); b = np.asarray(b, float)
    n = K["s"]
    for seed in range(6):
        rng = np.random.default_rng(seed)
        Y0 = rng.standard_normal((n, 1)); Y0 /= np.linalg.norm(Y0)
        opts = dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, Y0=Y0)
        t = time.time(); Yr, objr, dr = R.ManiSDP_unittrace(At, b, c, K, dict(opts), verbose=False); tr = time.time() - t
        line = "%s seed %d: oracle obj %.6f eta %.1e status %d iters %d (%.1f s)" % (
            case, seed, -objr, max(dr["gap"], dr["pinf"], dr["dinf"]), dr["status"], dr["iters"], tr)
        for mode in ("host", "device"):
            t = time.time(); Y, obj, d = solvers.ManiSDP_unittrace(At, b, c, K, dict(opts, eig=mode), verbose=False); tg = time.time() - t
            line += " | gpu(%s) obj %.6f eta %.1e status %d iters %d (%.1f s)" % (
                mode, -obj, max(d["gap"], d["pinf"], d["dinf"]), d["status"], d["iters"], tg)
        print(line + "   known %.5f" % known[case], flush=True)
