"""BQP d = 60 from the DEFAULT start point: which ingredient of the GPU path makes the AL trajectory stall
?  Variants: A(YaYb') route, host LAPACK eig(S) vs the device Lanczos escape,
host vs device AL bookkeeping, SVD vs Gram rank cut."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers

d = int(os.environ.get("BQP_D", "60"))
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()

variants = sys.argv[1:] or ["default", "host_eig", "sddmm", "svd_cut", "host_al"]
for v in variants:
    opts = {}
    os.environ.pop("MSDP_AFFINE_ROUTE", None)
    solvers._RANK_CUT_SVD = False
    if v == "host_eig":
        opts["eig"] = "host"; opts["dense_eig_max"] = 10 ** 9
    elif v == "sddmm":
        os.environ["MSDP_AFFINE_ROUTE"] = "sddmm"
    elif v == "svd_cut":
        solvers._RANK_CUT_SVD = True
    elif v == "host_al":
        opts["device_al"] = False
    elif v == "host_eig_svd":
        opts["eig"] = "host"; opts["dense_eig_max"] = 10 ** 9; solvers._RANK_CUT_SVD = True
    elif v.startswith("maxit"):
        opts["AL_maxiter"] = int(v[5:])
    t = time.time()
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, opts, verbose=bool(os.environ.get("BQP_VERBOSE")))
    print("variant %-12s obj %.8f eta %.1e status %d iters %d hessvecs %d p %d  %.1f s" % (
        v, obj, max(data["gap"], data["pinf"], data["dinf"]), data["status"], data["iters"], data["hessvecs"],
        Y.shape[1], time.time() - t), flush=True)
