"""GPU test of the RCCL data path on ONE GPU: a size-1 communicator routes every exchange through the same
ncclAllGather / ncclAllReduce calls the 8-GPU run uses (all-gather of the thin direction into the gather
buffer before S*U, all-reduce of the partial-sum arrays, chunked tCG without host-timing dependence).
Results must be bit-identical to the communicator-free path."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu


def test_size1_communicator_is_bit_identical(monkeypatch):
    monkeypatch.setenv("MSDP_NO_PERSIST", "1")       # compare with the communicator-free CHUNKED path (same kernels)
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n, p = C.shape[0], 12
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    res = []
    for use_comm in (False, True):
        h = _lib.Handle.onlyunitdiag(C)
        if use_comm:
            h.comm_init(1, 0, _lib.Handle.comm_unique_id())
            assert h.local_rows() == (0, n)
        h.set_point(Y)
        H = h.hessvec(U)
        G = h.rgrad()
        st = h.rtr(_lib.default_opts(maxiter=10, maxinner=30, tolgradnorm=1e-8))
        res.append((H, G, st.cost, st.gradnorm, st.hessvecs, st.accepted, h.get_point()))
        h.close()
    a, b = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and a[5] == b[5]
    assert np.array_equal(a[6], b[6])


@pytest.mark.parametrize("N", [2, 3, 8])
@pytest.mark.parametrize("case,p", [("G11", 12), ("G1", 7)])
def test_sparse_shards_match_unsharded(N, case, p):
    """Every row shard of the sparse path (rank r of N standing alone on one GPU; the test fills the gather buffer
    that the RCCL all-gather fills in the N-GPU run) reproduces its rows of the cost state, the gradient and the
    Hess-vec of the unsharded problem: local CSR/ELL rows with global column indices, row0 != 0, ragged last shard."""
    from manisdp_matlab_amd import _lib, problems
    from oracle import manisdp_ref as R
    _lib.load()
    C = problems.maxcut_cost_matrix(golden_path(case + ".txt.gz"))
    n = C.shape[0]
    rng = np.random.default_rng(N)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    z_ref = np.asarray(np.sum((C @ Y) * Y, axis=1)).ravel()
    covered = np.zeros(n, bool)
    for r in range(N):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.debug_shard(N, r)
        r0, r1 = h.local_rows()
        covered[r0:r1] = True
        h.set_point(Y)
        h.debug_set_full_rows(Y)
        G = h.rgrad()
        assert np.linalg.norm(G[r0:r1] - G_ref[r0:r1]) <= 1e-12 * np.linalg.norm(G_ref)
        assert np.linalg.norm(h.get_z()[r0:r1] - z_ref[r0:r1]) <= 1e-12 * np.linalg.norm(z_ref)
        h.debug_set_full_rows(U)
        H = h.hessvec(U)
        assert np.linalg.norm(H[r0:r1] - H_ref[r0:r1]) <= 1e-12 * np.linalg.norm(H_ref)
        assert h.tcg_path() == 0                      # shards never take the single-GPU persistent path
        h.close()
    assert covered.all()
