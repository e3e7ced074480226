"""Create / solve / destroy many handles of every kind and watch the free device memory (hipMemGetInfo through torch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manisdp_matlab_amd import _lib, problems, solvers
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20
C = problems.maxcut_cost_matrix(os.path.join(gold, "G11.txt.gz"))
At, b, c, K = problems.from_sdpa(os.path.join(gold, "gpp100.dat-s.gz"))
f0 = None
for rep in range(6):
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"eig": "device"}, verbose=False)
    Y, obj2, d2 = solvers.ManiSDP_unitdiag(At, b, c, K, dict(sigma0=1e-1, sigma_min=1e-1, tau1=1e-2, tau2=1e-1, TR_maxinner=40, TR_maxiter=6, eig="device"), verbose=False)
    Y, obj3, d3 = solvers.ManiSDP(At, b, c, K, {"AL_maxiter": 30, "eig": "device"}, verbose=False)
    f = free_mb()
    if f0 is None: f0 = f
    print("rep %d  obj %.6f %.6f  free %.0f MB (delta %.1f MB)" % (rep, obj, obj2, f, f - f0), flush=True)
assert abs(f - f0) < 64, "device memory leak"
print("no leak")
