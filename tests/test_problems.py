"""CPU tests of the instance readers / generators against facts stated in the reference tree."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_path
from manisdp_matlab_amd import problems as P


def test_gset_sizes():
    C = P.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    assert C.shape == (800, 800) and C.nnz == 39152            # K1
    C = P.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    assert C.shape == (20000, 20000) and C.nnz == 92644        # K2 (zero diagonals dropped)
    assert abs(C - C.T).max() == 0


def test_laplacian_assignment_semantics(tmp_path):
    """Laplacian.m:7-10: off-diagonals are ASSIGNED (last wins), the diagonal accumulates."""
    f = tmp_path / "g.txt"
    f.write_text("3 3\n1 2 1\n1 2 5\n2 3 2\n")
    L = P.gset_laplacian(str(f)).toarray()
    assert L[0, 1] == -5 and L[1, 0] == -5
    assert L[0, 0] == 6 and L[1, 1] == 8 and L[2, 2] == 2


def test_get_basis_order_matches_reference_iteration():
    for n, d in [(2, 2), (3, 2), (3, 4), (4, 3), (5, 4)]:
        assert (P.get_basis(n, d) == P._get_basis_literal(n, d)).all()
    # n=3, d=2 within degree 2: 200,110,020,101,011,002
    b = P.get_basis(3, 2)[:, 4:]
    assert [tuple(c) for c in b.T] == [(2, 0, 0), (1, 1, 0), (0, 2, 0), (1, 0, 1), (0, 1, 1), (0, 0, 2)]


def test_bqpmom_sizes_match_reference_log():
    """data/bqp_result.txt:3-8: d -> (n, m)."""
    for d, (n, m) in {10: (56, 1256), 20: (211, 16361)}.items():
        Q = np.loadtxt(golden_path(f"bqp_Q_{d}_1.txt.gz"), delimiter=",")
        e = np.loadtxt(golden_path(f"bqp_e_{d}_1.txt.gz"), delimiter=",")
        At, b, c, K = P.bqpmom(d, Q, e)
        assert K["s"] == n and At.shape == (n * n, m) and b[0] == 1 and np.count_nonzero(b) == 1
        # every constraint matrix is symmetric
        k = m // 2
        Ak = At[:, k].toarray().reshape(n, n, order="F")
        assert np.array_equal(Ak, Ak.T)
        # objective reproduces x'Qx + e'x on a +-1 point lifted to the moment matrix
        rng = np.random.default_rng(d)
        x = rng.choice([-1.0, 1.0], size=d)
        v = np.concatenate([[1.0], x, [x[j] * x[i] for i in range(1, d) for j in range(i)]])
        X = np.outer(v, v)
        C = c.toarray().reshape(n, n, order="F")
        assert abs(np.sum(C * X) - (x @ Q @ x + e @ x)) < 1e-9
        assert np.linalg.norm(At.T @ X.ravel(order="F") - b) < 1e-9      # feasible for every constraint


def test_qsmom_feasible_moment_matrix():
    coe = np.loadtxt(golden_path("qs_c_10_1.txt.gz"), delimiter=",")
    At, b, c, K = P.qsmom(10, coe)
    n = K["s"]
    assert n == 66 and At.shape[1] == n * (n + 1) // 2 - 1001 + n + 1
    rng = np.random.default_rng(0)
    x = rng.standard_normal(10); x /= np.linalg.norm(x)
    basis = P.get_basis(10, 2)
    v = np.prod(x[:, None] ** basis, axis=0)
    X = np.outer(v, v)
    assert np.linalg.norm(At.T @ X.ravel(order="F") - b) < 1e-9


def test_from_sdpa_structure():
    At, b, c, K = P.from_sdpa(golden_path("gpp100.dat-s.gz"))
    n = K["s"]
    assert n == 100 and At.shape == (10000, 101) and b[0] == 0 and (b[1:] == 1).all()
    A1 = At[:, 0].toarray().reshape(n, n, order="F")
    assert (A1 == 1).all()                                      # <J, X> = 0
    A2 = At[:, 1].toarray().reshape(n, n, order="F")
    assert A2[0, 0] == 1 and A2.sum() == 1                      # X_11 = 1
    At, b, c, K = P.from_sdpa(golden_path("theta1.dat-s.gz"))
    n = K["s"]
    A1 = At[:, 0].toarray().reshape(n, n, order="F")
    assert np.array_equal(A1, np.eye(n)) and b[0] == 1          # trace constraint
    assert (c.toarray() == -1).all()                            # F0 = J -> c = -vec(J)


def test_theta_generator():
    At, b, c, K = P.theta_problem(40, seed=1)
    n = K["s"]
    assert b[-1] == 1 and (b[:-1] == 0).all()
    last = At[:, -1].toarray().reshape(n, n, order="F")
    assert np.array_equal(last, np.eye(n))
    A0 = At[:, 0].toarray().reshape(n, n, order="F")
    assert np.array_equal(A0, A0.T) and A0.sum() == 2


def _sparse_moment_vector(cliques, x):
    vec = []
    for I in cliques:
        v = [1.0] + [x[a] for a in I]
        for jb in range(1, len(I)):
            for ia in range(jb):
                v.append(x[I[ia]] * x[I[jb]])
        v = np.array(v)
        vec.append(np.outer(v, v).ravel(order="F"))
    return np.concatenate(vec)


@pytest.mark.parametrize("t,q", [(1, 4), (3, 4), (4, 5), (3, 6)])
def test_bqpmom_sparse_structure(t, q):
    """Sparse second-order moment relaxation of a BQP (bqpmom_sparse.m): block sizes, the reference's constraint count
    (bqpmom_sparse.m:46), and validity -- the moments of every point of {-1,1}^n satisfy all constraints and reproduce f."""
    import itertools
    cl, n = P.chain_cliques(t, q)
    assert n == q + (q - 2) * (t - 1) and all(len(I) == q for I in cl) and cl[-1][-1] == n - 1
    mons = P.bqp_sparse_monomials(cl)
    rng = np.random.default_rng(10 * t + q)
    coe = rng.standard_normal(len(mons))
    At, b, c, K = P.bqpmom_sparse(n, cl, coe)
    mb = np.array(K["s"]); mc = np.array([len(I) for I in cl])
    assert K["nob"] == t and (mb == 1 + mc + mc * (mc - 1) // 2).all() and At.shape[0] == int(np.sum(mb * mb))
    support = set()                                            # monomials of degree <= 4, exponents <= 2, not a square
    for I in cl:
        for d in range(1, 5):
            for combo in itertools.combinations_with_replacement(I, d):
                cnt = [combo.count(v) for v in set(combo)]
                if max(cnt) <= 2 and any(e % 2 for e in cnt):
                    support.add(combo)
    assert At.shape[1] == int(np.sum(mb * (mb + 1) // 2) - len(support) + np.sum(mc * (mb - 1)) - np.sum(mb) + t)
    assert b[0] == 1 and not b[1:].any()
    for _ in range(4):
        x = rng.choice([-1.0, 1.0], n)
        X = _sparse_moment_vector(cl, x)
        f = sum(cv * np.prod([x[a] for a in mon]) for mon, cv in zip(mons, coe))
        assert np.abs(At.T @ X - b).max() == 0.0 and abs(c @ X - f) < 1e-12
    # symmetric constraint and cost matrices, block by block
    off = np.concatenate([[0], np.cumsum(mb * mb)])
    for k in range(t):
        Ck = c[off[k]:off[k + 1]].reshape(mb[k], mb[k])
        assert np.array_equal(Ck, Ck.T)
    col = At[:, At.shape[1] - 1].toarray().ravel()
    for k in range(t):
        Ak = col[off[k]:off[k + 1]].reshape(mb[k], mb[k])
        assert np.array_equal(Ak, Ak.T)


def test_bqpmom_sparse_relaxation_is_tight_on_a_small_chain():
    """Oracle ManiSDP_multiblock on the sparse relaxation (options of example_bqp_sparse.m:25-29) against brute force over
    {-1,1}^8: the relaxation of this chain is exact."""
    import itertools
    from oracle import manisdp_ref as R
    cl, n = P.chain_cliques(3, 4)
    mons = P.bqp_sparse_monomials(cl)
    coe = np.random.default_rng(1).standard_normal(len(mons))
    At, b, c, K = P.bqpmom_sparse(n, cl, coe)
    best = min(sum(cv * np.prod([x[a] for a in mon]) for mon, cv in zip(mons, coe)) for x in itertools.product([-1.0, 1.0], repeat=n))
    Y, obj, d = R.ManiSDP_multiblock(At, b, c, K, {"tol": 1e-8, "line_search": 1, "tau1": 1})
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    assert abs(obj - best) <= 1e-7 * abs(best)


def _chain_sphere_point(cliques, n, rng):
    """A point whose every clique sub-vector has unit norm (the feasible set of the sparse quartic problem)."""
    x = np.zeros(n); done = set()
    for I in cliques:
        free = [a for a in I if a not in done]
        s = sum(x[a] ** 2 for a in I if a in done)
        v = rng.standard_normal(len(free)); v *= np.sqrt(max(1.0 - s, 0.0)) / np.linalg.norm(v)
        x[free] = v
        done.update(I)
    return x


@pytest.mark.parametrize("t,q", [(1, 3), (2, 4), (3, 4), (3, 5)])
def test_qsmom_sparse_structure(t, q):
    """Sparse second-order moment relaxation of a quartic on clique spheres (qsmom_sparse.m): block sizes, the reference's
    constraint count (qsmom_sparse.m:29), validity on feasible points."""
    cl, n = P.chain_cliques(t, q)
    mons = P.quartic_sparse_monomials(cl)
    rng = np.random.default_rng(10 * t + q)
    coe = rng.standard_normal(len(mons))
    At, b, c, K = P.qsmom_sparse(n, cl, coe)
    mb = np.array(K["s"])
    assert K["nob"] == 0 and (mb == (q + 2) * (q + 1) // 2).all()
    assert At.shape == (int(np.sum(mb * mb)), int(np.sum(mb * (mb + 1) // 2) - len(mons) + np.sum(mb) + 1))
    for _ in range(3):
        x = _chain_sphere_point(cl, n, rng)
        assert all(abs(np.linalg.norm(x[I]) - 1.0) < 1e-12 for I in cl)
        vec = []
        for I in cl:
            v = [1.0] + [x[a] for a in I]
            for jb in range(len(I)):
                for ia in range(jb + 1):
                    v.append(x[I[ia]] * x[I[jb]])
            v = np.array(v)
            vec.append(np.outer(v, v).ravel(order="F"))
        X = np.concatenate(vec)
        f = sum(cv * np.prod([x[a] for a in mon]) for mon, cv in zip(mons, coe))
        assert np.abs(At.T @ X - b).max() < 1e-14 and abs(c @ X - f) < 1e-12


def test_qsmom_sparse_oracle_solve_is_a_lower_bound():
    from oracle import manisdp_ref as R
    cl, n = P.chain_cliques(3, 4)
    mons = P.quartic_sparse_monomials(cl)
    coe = np.random.default_rng(1).standard_normal(len(mons))
    At, b, c, K = P.qsmom_sparse(n, cl, coe)
    Y, obj, d = R.ManiSDP_multiblock(At, b, c, K, {"tol": 1e-8})
    assert d["status"] == 0 and max(d["gap"], d["pinf"], d["dinf"]) < 1e-8
    rng = np.random.default_rng(3)
    f = lambda x: sum(cv * np.prod([x[a] for a in mon]) for mon, cv in zip(mons, coe))      # noqa: E731
    assert all(obj <= f(_chain_sphere_point(cl, n, rng)) + 1e-8 for _ in range(200))


def test_synthetic_dense_generator_matches_the_library_host_function():
    """problems.SyntheticDenseC (NumPy) against msdp_synthetic_dense_entry, the host twin of the generator every rank runs on
    the device for its rows of BASELINE config 5's matrix (a pure host function: callable without a GPU)."""
    import ctypes
    import os
    from manisdp_matlab_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.msdp_synthetic_dense_entry.restype = ctypes.c_double
    lib.msdp_synthetic_dense_entry.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64]
    for n, seed in ((37, 0), (1000, 5), (100000, 0)):
        Cs = P.SyntheticDenseC(n, seed)
        rows = [0, 1, n // 3, n - 1]
        R = Cs.rows(rows)
        for q, i in enumerate(rows):
            for j in (0, 1, i, n // 2, n - 1):
                assert R[q, j] == lib.msdp_synthetic_dense_entry(n, i, j, seed)
                assert R[q, j] == lib.msdp_synthetic_dense_entry(n, j, i, seed)              # symmetric
        assert np.abs(R).max() <= 1.0 / np.sqrt(n)
    A = P.SyntheticDenseC(50, 2).toarray()
    assert np.array_equal(A, A.T)
    with pytest.raises(ValueError):
        P.SyntheticDenseC(30000).toarray()


def test_matrix_completion_generator_and_oracle():
    """example_matrixcompletion.m:8-41 restated: every constraint reads X[j, p+l] + X[p+l, j] = 2 M[j, l] on a sampled
    position, the cost is the trace; the lifted matrix [U; V][U; V]' of a factorisation M = U V' is feasible; the oracle's
    generic ManiSDP recovers M (tr X = 2 |M|_*)."""
    from oracle import manisdp_ref as R
    p, q, k = 30, 25, 2
    At, b, c, K, M, (j, l) = P.matrix_completion(p, q, k, m=6000, seed=3)
    n = p + q
    assert K["s"] == n and At.shape == (n * n, b.size) and np.array_equal(c.reshape(n, n), np.eye(n))
    assert len(set(zip(j.tolist(), l.tolist()))) == b.size and np.array_equal(b, 2.0 * M[j, l])
    A0 = At[:, 0].toarray().reshape(n, n, order="F")
    assert A0[j[0], p + l[0]] == 1 and A0[p + l[0], j[0]] == 1 and A0.sum() == 2
    U, s, Vt = np.linalg.svd(M, full_matrices=False)
    L = np.vstack([U[:, :k] * np.sqrt(s[:k]), Vt[:k].T * np.sqrt(s[:k])])
    X = L @ L.T
    assert np.abs(At.T @ X.ravel(order="F") - b).max() < 1e-12 and abs(np.trace(X) - 2 * s.sum()) < 1e-10
    Y, obj, d = R.ManiSDP(At, b, c, K, {"tol": 1e-8, "theta": 1e-2, "TR_maxinner": 6, "TR_maxiter": 8, "delta": 10, "alpha": 0.1})
    assert d["status"] == 0 and abs(obj - 2 * s.sum()) <= 1e-6 * obj
    assert np.linalg.norm(Y[:p] @ Y[p:].T - M) <= 1e-6 * np.linalg.norm(M)


def test_quasar_relaxation_is_valid_on_feasible_points():
    """problems.quasar_problem (the SDP of example/example_rotationsearch.m:26-28, whose generator belongs to STRIDE and is
    restated here from the QUASAR paper): for every unit quaternion q and every sign pattern theta the matrix Z = xx',
    x = [q; theta_1 q; ...], satisfies all constraints, and <C, Z> is the truncated-least-squares cost it stands for -- written
    out with the rotation matrix of q, which also pins the quaternion convention.  Constraint count: 1 + 10 N + 6 N + 6 N(N-1)/2."""
    N = 7
    a, b, R, beta, out = P.wahba_with_outliers(N, 0.5, seed=3)
    assert out.sum() == 4 and np.allclose(np.linalg.norm(a, axis=1), 1.0)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1.0) < 1e-14
    assert np.abs(b[~out] - a[~out] @ R.T).max() < 0.06                          # inliers: noise of sigma = 0.01
    At, bv, c, K = P.quasar_problem(a, b, beta ** 2)
    n = 4 * (N + 1)
    assert K["s"] == n and At.shape == (n * n, 1 + 16 * N + 3 * N * (N - 1))
    assert bv[0] == 1.0 and np.count_nonzero(bv) == 1
    C = c.reshape(n, n, order="F")
    assert np.array_equal(C, C.T)
    for k in (0, 5, 80, At.shape[1] - 1):                                        # symmetric constraint matrices
        Ak = np.asarray(At[:, k].todense()).reshape(n, n, order="F")
        assert np.array_equal(Ak, Ak.T)
    rng = np.random.default_rng(0)
    for _ in range(4):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        th = rng.choice([-1.0, 1.0], N)
        x = np.concatenate([q] + [t * q for t in th])
        z = np.outer(x, x).reshape(-1, order="F")
        assert np.abs(At.T @ z - bv).max() < 1e-14
        Rq = P._quat_rotation(q)
        direct = sum((1 + t) / 2 * np.sum((bi - Rq @ ai) ** 2) / beta ** 2 + (1 - t) / 2 for ai, bi, t in zip(a, b, th))
        assert abs(c @ z - direct) <= 1e-12 * abs(direct)
    At2 = P.quasar_problem(a, b, beta ** 2, redundant=False)[0]
    assert At2.shape[1] == 1 + 16 * N


def test_snl_relaxation_is_valid():
    """problems.snl_polynomial / snl_mom (src/basicfunction/snl_mom_sparse.m:4-96 for one clique, the workload of
    example/Sensor_Network_Localization.m:2-35): the quartic vanishes at the true sensor positions, the moments of ANY point satisfy
    every constraint and reproduce f, and the sizes are the reference's (mb = C(2n+2, 2), columns = mb (mb + 1) / 2 - C(2n+4, 4) + mb + 1)."""
    from math import comb
    from manisdp_matlab_amd import problems
    n = 4
    f, loc = problems.snl_polynomial(n, seed=1)
    At, b, c, K = problems.snl_mom(f, 2 * n)
    mb = K["s"]
    assert mb == comb(2 * n + 2, 2) and At.shape == (mb * mb, mb * (mb + 1) // 2 - comb(2 * n + 4, 4) + mb + 1)
    ev = lambda x: sum(cf * np.prod([x[i] for i in m]) for m, cf in f.items())
    assert abs(ev(np.concatenate([loc[0], loc[1]]))) < 1e-12
    ba = problems.get_basis(2 * n, 2)
    rng = np.random.default_rng(0)
    for _ in range(3):
        x = rng.standard_normal(2 * n)
        v = np.array([np.prod(x ** ba[:, k]) for k in range(mb)])
        X = np.outer(v, v).reshape(-1, order="F")
        assert np.abs(At.T @ X - b).max() < 1e-12 * max(1.0, np.abs(X).max())
        assert abs((c.T @ X)[0] - ev(x)) <= 1e-10 * max(1.0, abs(ev(x)))
    C = np.asarray(c.todense()).reshape(mb, mb)
    assert np.array_equal(C, C.T)
