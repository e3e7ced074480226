"""Theta function of a Hamming graph with ManiSDP_unittrace: the SeDuMi data of the reference's example/generate_hamming.m (the
generator behind SDPLIB's hamming_* problems) through the unit-trace entry point.  argv = [k, default 7] [distances, default 5,6]
-- H_{7,{5,6}} is SDPLIB's hamming_7_5_6 (theta = 128/3 = 42.6667)."""
import sys
import time

from _common import eta
from manisdp_matlab_amd import problems, solvers

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
d = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5, 6]
At, b, c, K = problems.generate_hamming(k, d)
t = time.time()
Y, fval, data = solvers.ManiSDP_unittrace(At, b, c, K, {"tol": 1e-8, "TR_maxiter": 30, "TR_maxinner": 200})
print("H_{%d,%s}: n = %d, m = %d" % (k, d, K["s"], At.shape[1]))
print("ManiSDP: theta = %.8f, eta = %.1e, status = %d, time = %.2fs" % (-fval, eta(data), data["status"], time.time() - t))
