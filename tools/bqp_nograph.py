"""BQP d = 60, start point 1: RTR seconds with and without hipGraph chunks (MSDP_NO_GRAPH)."""
import os, sys, time, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from manisdp_matlab_amd import problems, solvers
    d = 60
    gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
    e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
    At, b, c, K = problems.bqpmom(d, Q, e)
    c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
    t = time.time()
    Y, obj, data = solvers.ManiSDP_unitdiag(At, b, c, K, {}, verbose=False, rng=np.random.default_rng(1))
    tt = time.time() - t
    print("NO_GRAPH=%s: obj %.8f iters %d hessvecs %d  %.2f s (rtr %.2f s, eig %.2f s, host %.2f s)" % (
        os.environ.get("MSDP_NO_GRAPH", "0"), obj, data["iters"], data["hessvecs"], tt, data["rtr_seconds"], data["eig_seconds"],
        tt - data["rtr_seconds"] - data["eig_seconds"]), flush=True)
else:
    for v in ("0", "1"):
        env = dict(os.environ); env["MSDP_NO_GRAPH"] = v
        if v == "0": env.pop("MSDP_NO_GRAPH")
        subprocess.run([sys.executable, __file__, "run"], env=env, check=False)
