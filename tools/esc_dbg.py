import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from manisdp_matlab_amd import _lib, problems
C = problems.maxcut_cost_matrix('/root/repo/tests/golden/G1.txt.gz')
rng = np.random.default_rng(0)
for p in (2,10):
    Y = rng.standard_normal((C.shape[0], p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C); h.set_point(Y)
    st = h.rtr(_lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
    z = h.get_z(); S = C.toarray() - np.diag(z); dS, vS = np.linalg.eigh(S)
    for tol in (1e-6, 1e-9, 1e-12):
        lam, V, lmax, its = h.escape_eigs(8, tol=tol, maxit=600)
        res = [np.linalg.norm(S@V[:,t]-lam[t]*V[:,t]) for t in range(8)]
        print(p, 'gradnorm', st.gradnorm, 'tol', tol, 'its', its, 'lam', lam[:4], 'ref', dS[:4], 'res', np.array(res).round(9), 'lmax', lmax, dS[-1])
    h.close()
