"""A few plain launches of the chunked tCG kernels at a large sparse size (toroidal rows x cols grid) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSDP_NO_GRAPH"] = "1"
import numpy as np
from manisdp_matlab_amd import _lib, problems
rows, cols, p = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 1000, 32)
sweep = int(sys.argv[4]) if len(sys.argv) > 4 else None
C = problems.toroidal_grid_maxcut(rows, cols, seed=3)
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
if sweep is not None:
    h.set_option("sweep", sweep)
h.set_point(Y)
ms, by, fl = h.bench_hessvec(8)
print("n", n, "p", p, "hess us", ms * 1e3, "algorithmic bytes", by, "trip us", h.bench_tcg_trip(8) * 1e3, "path", h.tcg_path())
