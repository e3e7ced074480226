"""Row sharding of the affine kinds on ONE GPU:
 * a size-1 communicator runs the code path of the N-GPU run (gathered copies of the point, row offsets into the
   replicated operator state, all-reduced partial sums, replicated escape) and must give bit-identical results to the
   communicator-free handle -- operators, a trustregions() call, the AL bookkeeping, the escape, whole solves;
 * every row shard (rank r of N standing alone; the test fills the gather buffer the RCCL all-gather fills in the N-GPU
   run) reproduces its rows of the gradient and of the Hess-vec of the unsharded problem.
N processes on N GPUs: tests/test_gpu_comm.py (auto-skipped on the single-GPU box)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_path

pytestmark = pytest.mark.gpu


def _problem(kind_name):
    from manisdp_matlab_amd import _lib, problems
    if kind_name == "unitdiag":
        Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
        e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
        At, b, c, K = problems.bqpmom(10, Q, e)
        kind = _lib.KIND_UNITDIAG
    else:
        At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
        kind = _lib.KIND_UNITTRACE if kind_name == "unittrace" else _lib.KIND_GENERIC
    c = np.asarray(c.todense()).ravel() if sp.issparse(c) else np.asarray(c, float).ravel()
    b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    At = sp.csc_matrix(At); At.sort_indices()
    return kind, At, b, c, K


def _point(kind, n, p, rng):
    from manisdp_matlab_amd import _lib
    Y = rng.standard_normal((n, p))
    if kind == _lib.KIND_UNITDIAG:
        Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    elif kind == _lib.KIND_UNITTRACE:
        Y /= np.linalg.norm(Y)
    return Y


@pytest.mark.parametrize("kind_name", ["unitdiag", "unittrace", "generic"])
def test_size1_communicator_affine_is_bit_identical(kind_name):
    from manisdp_matlab_amd import _lib
    _lib.load()
    kind, At, b, c, K = _problem(kind_name)
    n, m, p = K["s"], b.size, 6
    rng = np.random.default_rng(3)
    Y = _point(kind, n, p, rng)
    U = 0.3 * rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(m)
    res = []
    for use_comm in (False, True):
        h = _lib.Handle.affine(kind, At, b, c, n)
        if use_comm:
            h.comm_init(1, 0, _lib.Handle.comm_unique_id())
        else:
            # round 4: a communicator-free handle takes the B route / the fused SDDMM (msdp_affine.hip), which a row-sharded one does
            # not -- switched off here so that the comparison stays one of the communicator plumbing, bit for bit
            h.set_option("affine_broute", 0); h.set_option("affine_fuse", 0)
        h.set_multipliers(y, 0.7)
        h.set_point(Y)
        f, G, H = h.cost(), h.rgrad(), h.hessvec(h.proj(U))
        co = (h.linesearch_cost(None, 0.0), h.linesearch_cost(U, 0.5))
        st = h.rtr(_lib.default_opts(maxiter=3, maxinner=15, tolgradnorm=1e-8))
        Yr = h.get_point_all()
        obj, Ax = h.al_primal(m)
        z = h.al_dual(y)
        lam, V, lmax, _ = h.escape_eigs_dual(3, tol=1e-10, maxit=4000)
        S = h.get_dual_slack()
        res.append((f, G, H, co, st.cost, st.gradnorm, st.hessvecs, Yr, obj, Ax, np.atleast_1d(np.asarray(0.0 if z is None else z, float)), lam, V, lmax, S))
        h.close()
    for k, (a, bb) in enumerate(zip(*res)):
        a, bb = np.asarray(a, float), np.asarray(bb, float)
        if kind_name != "unitdiag":
            # theta1's At touches few entries of the matrix: one GPU takes the restricted-adjoint / sparse-product route
            # (DESIGN.md section 4), the row-sharded path the dense one -- the same numbers in another order; the global
            # norms of the sphere go through the all-reduced partial array likewise
            if k == 11:                                   # missing pairs come back as +inf
                assert np.array_equal(np.isfinite(a), np.isfinite(bb))
                a, bb = a[np.isfinite(a)], bb[np.isfinite(bb)]
            if k == 12:                                   # eigenvectors of the valid pairs: up to sign
                ok = np.isfinite(np.asarray(res[0][11], float))
                a, bb = a[:, ok], bb[:, ok]
                bb = bb * np.sign(np.sum(a * bb, axis=0))
            assert np.linalg.norm(a - bb) <= 1e-9 * max(1.0, np.linalg.norm(a)), k
        else:
            assert np.array_equal(a, bb), k


@pytest.mark.parametrize("kind_name", ["unitdiag", "unittrace"])
def test_size1_communicator_full_solve(kind_name):
    """The host loop with options['comm']: same iterates as the communicator-free solve."""
    from manisdp_matlab_amd import _lib, solvers
    kind, At, b, c, K = _problem(kind_name)
    n = K["s"]
    solve = solvers.ManiSDP_unitdiag if kind_name == "unitdiag" else solvers.ManiSDP_unittrace
    rng = np.random.default_rng(5)
    Y0 = _point(kind, n, 2 if kind_name == "unitdiag" else 1, rng)
    o = {"Y0": Y0, "tol": 1e-8, "AL_maxiter": 60}
    # the communicator-free handle keeps the operator routes a row-sharded one takes (round 4's B route / fused SDDMM are one-rank
    # forms): the comparison is one of the communicator plumbing, iterate for iterate
    Ya, obja, da = solve(At, b, c, K, dict(o, device_options={"affine_broute": 0, "affine_fuse": 0}), verbose=False)
    Yb, objb, db = solve(At, b, c, K, dict(o, comm=(1, 0, _lib.Handle.comm_unique_id())), verbose=False)
    if kind_name == "unitdiag":
        assert da["iters"] == db["iters"] and da["hessvecs"] == db["hessvecs"]
        assert obja == objb and np.array_equal(Ya, Yb)
    else:
        # sphere: the global norms take another summation order (see above) and theta1's AL trajectory is sensitive to
        # the last bits: the first outer iterates must agree, the end points need not
        for k in range(4):
            ga, gb = da["log"][k], db["log"][k]
            assert abs(ga[1] - gb[1]) <= 1e-7 * max(1.0, abs(ga[1])), (k, ga, gb)      # obj
            assert abs(ga[3] - gb[3]) <= 1e-6 * max(1e-3, abs(ga[3])), (k, ga, gb)     # pinf
            assert ga[7] == gb[7]                                                      # p


@pytest.mark.parametrize("N", [2, 3])
@pytest.mark.parametrize("route", [1, 2])
def test_unitdiag_shards_match_unsharded(N, route):
    """Gradient and Hess-vec rows of every shard against the unsharded handle (row offsets into the replicated eS / AyU,
    local row dots, ragged last shard), on both routes of the A(.) operator."""
    from manisdp_matlab_amd import _lib
    _lib.load()
    kind, At, b, c, K = _problem("unitdiag")
    n, m, p = K["s"], b.size, 5
    rng = np.random.default_rng(N)
    Y = _point(kind, n, p, rng)
    U = rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(m)
    h0 = _lib.Handle.affine(kind, At, b, c, n)
    h0.set_option("affine_route", route)
    h0.set_multipliers(y, 0.7)
    h0.set_point(Y)
    G_ref = h0.rgrad()
    Ut = h0.proj(U)
    H_ref = h0.hessvec(Ut)
    h0.close()
    covered = np.zeros(n, bool)
    for r in range(N):
        h = _lib.Handle.affine(kind, At, b, c, n)
        h.set_option("affine_route", route)
        h.debug_shard(N, r)
        r0, r1 = h.local_rows()
        covered[r0:r1] = True
        h.set_multipliers(y, 0.7)
        h.set_point(Y)
        h.debug_set_full_rows(Y)
        G = h.rgrad()
        assert np.linalg.norm(G[r0:r1] - G_ref[r0:r1]) <= 1e-12 * np.linalg.norm(G_ref)
        h.debug_set_full_rows(Ut)
        H = h.hessvec(Ut)
        assert np.linalg.norm(H[r0:r1] - H_ref[r0:r1]) <= 1e-12 * np.linalg.norm(H_ref)
        h.close()
    assert covered.all()
