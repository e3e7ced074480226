"""Few plain launches of each tCG kernel (no graphs) for rocprofv3 --pmc passes.  argv: [p = 32] [persist_pipe = 1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSDP_NO_GRAPH"] = "1"
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_option("persist_pipe", int(sys.argv[2]) if len(sys.argv) > 2 else 1)
h.set_point(Y)
trips = 64
print("trip us", h.bench_tcg_trip(trips) * 1e3, "path", h.tcg_path(), "trips per persistent launch", trips)
print("hess us", h.bench_hessvec(8)[0] * 1e3)
