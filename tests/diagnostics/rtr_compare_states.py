"""One trustregions() call on the GPU from states of the ORACLE's BQP d = 60 trajectory (tmp_states/*.npz, written by a
dump run of oracle/manisdp_ref.py: the point Y, multipliers y and sigma going INTO the RTR call of AL iteration k, and
what the oracle's RTR returned): does the device RTR do the same job on identical input?"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
d = 60
gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
Q = np.loadtxt(os.path.join(gold, "bqp_Q_%d_1.txt.gz" % d), delimiter=",")
e = np.loadtxt(os.path.join(gold, "bqp_e_%d_1.txt.gz" % d), delimiter=",")
At, b, c, K = problems.bqpmom(d, Q, e)
c = np.asarray(c.todense()).ravel(); c = c / np.abs(c).max()
n = K["s"]
h = _lib.Handle.affine(_lib.KIND_UNITDIAG, At, b, c, n, pcap=64)
files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tmp_states", "bqp60_it*.npz")),
               key=lambda f: int(f.split("_it")[1].split(".")[0]))
for f in files:
    S = np.load(f)
    Y, y, sigma = S["Y"], S["y"], float(S["sigma"])
    h.set_multipliers(y, sigma)
    h.set_point(Y)
    f0 = h.cost()
    st = h.rtr(_lib.default_opts(maxiter=4, maxinner=20, tolgradnorm=1e-8))
    Yg = h.get_point()
    print("%s p=%d sigma=%.3g  f0 %.10f | oracle: cost %.10f gradnorm %.3e hessvecs %d rejected %d | gpu: cost %.10f gradnorm %.3e hessvecs %d rejected %d | rel |Y_gpu - Y_oracle| %.2e"
          % (os.path.basename(f), Y.shape[1], sigma, f0, float(S["cost"]), float(S["gradnorm"]), int(S["hessvecs"]), int(S["rejected"]),
             st.cost, st.gradnorm, st.hessvecs, st.rejected, np.linalg.norm(Yg - S["Yout"]) / np.linalg.norm(S["Yout"])), flush=True)
h.close()
