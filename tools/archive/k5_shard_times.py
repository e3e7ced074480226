"""Per-GPU kernel rate of BASELINE config 5 (synthetic dense C, n = 100000, p = 64, rows over 8 GPUs): one
process stands in for rank 0 of 8 (12 500 x 100 000 slab = 10 GB generated on the device) and times the
S*U Hess-vec (MFMA kernel + epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
h = _lib.Handle.dense_synthetic(n, 0, nranks=N, rank=0, pcap=p)
r0, r1 = h.local_rows()
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h.set_point(Y)
h.debug_set_full_rows(Y)
for _ in range(2):
    ms, by, fl = h.bench_hessvec(50)
print("K5 shard: rows %d..%d of n=%d, p=%d: Hess-vec %.1f us, %.0f GB/s (%.1f%% of 8 TB/s), %.1f TFLOP/s fp64 (%.1f%% of 78.6)" % (
    r0, r1, n, p, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80, fl / ms / 1e9, fl / ms / 1e9 / 0.786))
