import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.sparse as sp
from conftest import golden_path, within_print
from manisdp_matlab_amd import problems, solvers
PRINTED = json.load(open(golden_path("known_answers_printed.json")))
THETA_OPTS = dict(tol=1e-8, TR_maxiter=30, TR_maxinner=200)
GPP_OPTS = dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6)
def sdpa(name):
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
    b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float).ravel()
    return At, b, c, K
for g, name in (("G55", "maxG55"), ("G60", "maxG60")):
    C = problems.maxcut_cost_matrix(golden_path(g + ".txt.gz"))
    for eig in ("device",):
        t = time.time()
        Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": eig}, verbose=False)
        print(name, eig, "obj %.6f printed %s ok=%s status %d dinf %.1e iters %d %.1fs" % (-obj, PRINTED[name], within_print(-obj, PRINTED[name]), data["status"], data["dinf"], data["iters"], time.time() - t), flush=True)
for name in ("gpp250-2", "gpp250-3", "gpp250-4", "gpp500-1", "gpp500-2", "gpp500-3", "gpp500-4"):
    At, b, c, K = sdpa(name)
    t = time.time()
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, dict(GPP_OPTS), verbose=False)
    print(name, "obj %.6f printed %s ok=%s status %d res %.1e iters %d %.1fs" % (-obj, PRINTED[name], within_print(-obj, PRINTED[name]), data["status"], max(data["gap"], data["pinf"], data["dinf"]), data["iters"], time.time() - t), flush=True)
for name in ("theta5", "theta6"):
    At, b, c, K = sdpa(name)
    for eig in ("host", "device"):
        t = time.time()
        Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, dict(THETA_OPTS, eig=eig), verbose=False)
        print(name, eig, "obj %.7f printed %s ok=%s status %d res %.1e iters %d %.1fs" % (-obj, PRINTED[name], within_print(-obj, PRINTED[name]), data["status"], max(data["gap"], data["pinf"], data["dinf"]), data["iters"], time.time() - t), flush=True)
