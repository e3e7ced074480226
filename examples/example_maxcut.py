"""MaxCut SDP relaxation of a Gset graph with ManiSDP_onlyunitdiag -- the call of the reference's
example/example_maxcut.m:9-34 (C = -L/4, options.p0 = 40, tol = 1e-8): argv = [graph name, default G81]."""
import sys
import time

from _common import GOLDEN, eta
from manisdp_matlab_amd import problems, solvers

name = sys.argv[1] if len(sys.argv) > 1 else "G81"
C = problems.maxcut_cost_matrix("%s/%s.txt.gz" % (GOLDEN, name))
t = time.time()
Y, fval, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40, "tol": 1e-8})
print("ManiSDP: optimum = %.8f, eta = %.1e, time = %.2fs (rank %d)" % (fval, eta(data), time.time() - t, Y.shape[1]))
