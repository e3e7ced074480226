"""Dense-C Hess-vecs on the synthetic generator for a rocprofv3 run: argv = n p [p ...] [--shard N] (rank 0 of N rows only) [--row-major]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib
args = sys.argv[1:]
N = 1
pack = 1
if "--row-major" in args:
    args.remove("--row-major"); pack = 0
if "--shard" in args:
    i = args.index("--shard"); N = int(args[i + 1]); del args[i:i + 2]
n = int(args[0])
for p in [int(x) for x in args[1:]]:
    h = _lib.Handle.dense_synthetic(n, 0, nranks=N, rank=0, pcap=p)
    h.set_option("dense_pack", pack)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h.set_point(Y)
    if N > 1:
        h.debug_set_full_rows(Y)
    # the first ~40 launches of a cold GPU run 5-35 % slower (684 -> 863 -> 640 us at n = 20000, p = 32: clock / power
    # transient), so the kernel-stat average is taken over a run long enough to be dominated by the steady state
    reps = 300 if n * (n // N) <= 4e8 else 100
    h.bench_hessvec(50)
    ms, by, fl = h.bench_hessvec(reps)
    print("n=%d p=%d shard 1/%d: Hess-vec %.1f us, %.0f GB/s algorithmic, %.1f TFLOP/s fp64" % (n, p, N, ms * 1e3, by / ms / 1e6, fl / ms / 1e9), flush=True)
    h.close()
