"""Hess-vec of the multiblock kind (per-block storage) on the sparse-BQP chain of t cliques of q variables: per-kernel times under
rocprofv3 --kernel-trace --stats.  argv: [t=100] [q=20] [p=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import _lib, problems
t = int(sys.argv[1]) if len(sys.argv) > 1 else 100
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cl, n = problems.chain_cliques(t, q)
coe = np.random.default_rng(1).standard_normal(len(problems.bqp_sparse_monomials(cl)))
At, b, c, K = problems.bqpmom_sparse(n, cl, coe)
nset = [int(v) for v in K["s"]]
N = sum(nset)
h = _lib.Handle.multiblock(At, b, c, nset, len(nset), pcap=max(32, p))
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h.set_option("graph", 0)
h.set_multipliers(0.1 * rng.standard_normal(b.size), 1.0)
h.set_point(Y)
h.cost(); h.rgrad()
U = h.proj(rng.standard_normal((N, p)))
for _ in range(30):
    h.hessvec(U)
print("done", N, b.size, At.nnz)
h.close()
