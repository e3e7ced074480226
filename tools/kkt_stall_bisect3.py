"""40-MB arrays are always mmap'ed by glibc (above its 32-MB ceiling of the dynamic threshold): allocate / free them around pieces of a
solve and time a small handle's upload after each step."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from manisdp_matlab_amd import _lib, problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G81.txt.gz"))
ns = 4000
Cs = sp.random(ns, ns, density=4.0 / ns, random_state=1, format="csr"); Cs = (Cs + Cs.T).tocsr()
hs = _lib.Handle.onlyunitdiag(Cs, pcap=40)
Ys = np.random.default_rng(2).standard_normal((ns, 40)); Ys /= np.linalg.norm(Ys, axis=1, keepdims=True)
def probe(what):
    t = time.perf_counter(); hs.set_point(Ys); t1 = time.perf_counter(); hs.cost(); dt = time.perf_counter() - t
    print("%-50s probe %.2f ms (set_point %.2f)" % (what, 1e3 * dt, 1e3 * (t1 - t)), flush=True)
def cycle(tag, reps=2):
    for rep in range(reps):
        big = np.ones(5_000_000); probe(tag + ": allocated 40 MB")
        del big; probe(tag + ": freed it"); probe(tag + ":   again")
probe("cold"); probe("warm")
cycle("before")
mode = sys.argv[1] if len(sys.argv) > 1 else "solve"
if mode == "solve":
    r = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
elif mode == "solve1":
    r = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40, "AL_maxiter": 1}, verbose=False)
elif mode == "rtr":
    h = _lib.Handle.onlyunitdiag(C.tocsr(), pcap=56)
    Y0 = np.random.default_rng(0).standard_normal((C.shape[0], 40)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    h.set_point(Y0); h.rtr(_lib.default_opts(maxiter=4, maxinner=100, tolgradnorm=1e-8)); r = h.get_point(); h.close()
elif mode == "create":
    h = _lib.Handle.onlyunitdiag(C.tocsr(), pcap=56); h.close(); r = None
probe("after " + mode)
cycle("after " + mode)
r = None; gc.collect()
probe("results freed")
cycle("results freed", 3)
