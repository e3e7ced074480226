"""A few fused trustregions() launches of bench.py's step (G81, seed-0 start, maxiter = 40, maxinner = 100) for rocprofv3 passes:
the dominant kernel of the bench line is ONE launch of k_tcg_pipe_obl<.., FUSE> per call.  argv: [p = 32] [calls = 4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSDP_NO_GRAPH"] = "1"
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
h.set_point(Y)
h.point_snapshot()
for _ in range(calls):
    h.point_restore()
    st = h.rtr(opts)
    print("call: %d Hess-vecs, %d iterations, device %.3f ms, path %d form %d" % (st.hessvecs, st.iters, h.last_rtr_device_ms(), h.tcg_path(), h.persist_form()))
h.close()
