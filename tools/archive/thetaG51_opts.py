"""thetaG51 (data/sdplib/README:105, n = 1001, m = 6910, printed optimum 3.49000e+02) through ManiSDP_unitdiag on the GPU
under a few option sets: which of them reach the printed digits.  (The oracle with the defaults stops at status 1 after
300 s: -349.0077, pinf 1.4e-4.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manisdp_matlab_amd import problems, solvers

At, b, c, K = problems.from_sdpa(os.path.join(ROOT, "tests/golden/thetaG51.dat-s.gz"))
c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float).ravel()
SETS = {
    "gpp_al60": dict(tol=1e-6, sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6, AL_maxiter=60),
    "gpp_al120": dict(tol=1e-6, sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6, AL_maxiter=120),
    "b10x200": dict(tol=1e-8, TR_maxiter=10, TR_maxinner=200),
    "b30x100": dict(tol=1e-8, TR_maxiter=30, TR_maxinner=100),
    "b30x200s": dict(tol=1e-8, TR_maxiter=30, TR_maxinner=200, sigma0=1e-1, sigma_min=1e-1),
    "b30x200t": dict(tol=1e-8, TR_maxiter=30, TR_maxinner=200, tau1=1e-1, tau2=1e-1),
    "b60x400": dict(tol=1e-8, TR_maxiter=60, TR_maxinner=400),
    "gpp6": dict(tol=1e-6, sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6),
    "default": dict(tol=1e-8),
    "gpp": dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6),
    "budget": dict(tol=1e-8, TR_maxiter=30, TR_maxinner=200),
    "budget_al": dict(tol=1e-8, TR_maxiter=10, TR_maxinner=100, AL_maxiter=1000),
    "sigma1": dict(tol=1e-8, sigma0=1.0, sigma_min=1.0, TR_maxiter=10, TR_maxinner=100, AL_maxiter=600),
    "ls": dict(tol=1e-8, line_search=1, TR_maxiter=10, TR_maxinner=100, AL_maxiter=600),
}
which = sys.argv[1:] or list(SETS)
for name in which:
    for eig in ("host",):
        t = time.time()
        try:
            Y, obj, d = solvers.ManiSDP_unitdiag(At, b, c, K, dict(SETS[name], eig=eig), verbose=False)
            print(f"{name:10s} eig={eig}: obj {-obj:.7f} status {d['status']} gap {d['gap']:.1e} pinf {d['pinf']:.1e} dinf {d['dinf']:.1e} "
                  f"{time.time() - t:.1f}s iters {d.get('iters')} hessvecs {d.get('hessvecs')} rtr {d.get('rtr_seconds', 0):.1f}s eig {d.get('eig_seconds', 0):.1f}s", flush=True)
        except Exception as ex:
            print(name, eig, "failed:", ex, flush=True)
