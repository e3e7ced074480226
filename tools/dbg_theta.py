import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
At, b, c, K = problems.from_sdpa(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "theta1.dat-s.gz"))
opts = dict(tol=1e-6, sigma0=1e5, sigma_max=1e8, AL_maxiter=int(sys.argv[1]) if len(sys.argv) > 1 else 12)
Y, obj, data = solvers.ManiSDP_unittrace(At, b, c, K, opts, verbose=True)
