"""GPU parity tests for the dense-C (fp64 MFMA) onlyunitdiag path, through the C-ABI,
against the oracle on seeded inputs.  Tolerance 1e-12 relative on operator outputs
(fp64; the MFMA k-order differs from NumPy's BLAS)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


@pytest.mark.parametrize("n,p", [(17, 1), (33, 5), (64, 2), (100, 3), (257, 16), (500, 31), (1000, 32), (1000, 48), (777, 64), (150, 80), (300, 100), (300, 130), (200, 260), (2050, 20)])
def test_dense_operators_match_oracle(lib, n, p):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    C = problems.dense_unitdiag_cost(n, seed=n)
    # asymmetric check of the MFMA C/D mapping: use a non-symmetric-looking but symmetric C with distinct rows
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f_ref = prob.cost(Y)
    assert abs(h.cost() - f_ref) <= 1e-12 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-12
    assert _relerr(h.hessvec(U), prob.hess(Y, U)) < 1e-12
    assert _relerr(h.get_z(), np.sum((C @ Y) * Y, axis=1)) < 1e-12
    h.close()


@pytest.mark.parametrize("n,p", [(33, 5), (1000, 32), (777, 64), (300, 130), (2050, 20)])
def test_dense_fragment_ordered_copy_is_bit_identical(lib, n, p):
    """The fragment-ordered copy of C feeds the same values into the same MFMA sequence as the row-major one."""
    from manisdp_matlab_amd import problems
    C = problems.dense_unitdiag_cost(n, seed=n + 1)
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    out = []
    for pack in (1, 0):
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("dense_pack", pack)
        h.set_point(Y)
        out.append((h.cost(), h.rgrad(), h.hessvec(U)))
        h.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


def test_dense_matches_sparse_path(lib):
    """The same matrix through the CSR kernels and through the MFMA kernels."""
    import scipy.sparse as sp
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(16, 25, seed=4)
    n, p = C.shape[0], 12
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    hs = lib.Handle.onlyunitdiag(C)
    hd = lib.Handle.onlyunitdiag(C.toarray())
    hs.set_point(Y); hd.set_point(Y)
    assert _relerr(hd.hessvec(U), hs.hessvec(U)) < 1e-13
    assert _relerr(hd.rgrad(), hs.rgrad()) < 1e-13
    hs.close(); hd.close()


def test_dense_solver_known_answer(lib):
    """mcp124-1 (SDPLIB value shipped with the reference) solved through the dense-C path."""
    import json
    from conftest import golden_path
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    At, b, c, K = problems.from_sdpa(golden_path("mcp124-1.dat-s.gz"))
    n = K["s"]
    C = c.toarray().reshape(n, n, order="F")
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {}, verbose=False)
    assert data["dinf"] < 1e-8
    assert abs(-obj - known["mcp124-1"]) < 1e-6 * abs(known["mcp124-1"])


def test_synthetic_dense_shards_match_full_matrix(lib):
    """Config-5 style pre-sharded dense C: every shard (rank r of 3, standing alone on one GPU with the gather
    buffer filled by the test) reproduces its rows of cost state, gradient and Hess-vec of the full problem."""
    from oracle import manisdp_ref as R
    n, p, seed, N = 301, 6, 7, 3
    L = lib.load()
    C = np.array([[L.msdp_synthetic_dense_entry(n, i, j, seed) for j in range(n)] for i in range(n)])
    assert np.array_equal(C, C.T)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f_ref = prob.cost(Y); G_ref = prob.grad(Y); H_ref = prob.hess(Y, U)
    # unsharded synthetic handle == dense handle built from the host copy of the same matrix
    h = lib.Handle.dense_synthetic(n, seed)
    h.set_point(Y)
    assert _relerr(h.rgrad(), G_ref) < 1e-12 and _relerr(h.hessvec(U), H_ref) < 1e-12
    h.close()
    f_sum = 0.0
    for r in range(N):
        h = lib.Handle.dense_synthetic(n, seed, nranks=N, rank=r)
        r0, r1 = h.local_rows()
        h.set_point(Y)
        h.debug_set_full_rows(Y)
        G = h.rgrad()
        assert _relerr(G[r0:r1], G_ref[r0:r1]) < 1e-12
        assert _relerr(h.get_z()[r0:r1], np.sum((C @ Y) * Y, axis=1)[r0:r1]) < 1e-12
        h.debug_set_full_rows(U)
        H = h.hessvec(U)
        assert _relerr(H[r0:r1], H_ref[r0:r1]) < 1e-12
        h.close()


@pytest.mark.parametrize("n,p,shape", [(17, 1, 1), (33, 5, 2), (130, 2, 3), (257, 16, 1), (300, 17, 2), (500, 31, 3), (1000, 32, 1),
                                       (1000, 32, 2), (1000, 32, 3), (2050, 20, 1), (2050, 7, 2), (3000, 32, 0)])
def test_symmetric_contraction_matches_oracle_and_full_kernel(lib, n, p, shape):
    """msdp_densesym.hip (upper triangle only, direct + transposed MFMA product per tile) against the oracle and against the
    full kernel of msdp_dense.hip, for the three workgroup shapes (dense_sym_rt: 8 x 16, 8 x 32, 16 x 16 rows; 0 = planned) and
    sizes around the tile / row-block edges; two runs give the same bits (no atomics, fixed summation orders)."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    C = problems.dense_unitdiag_cost(n, seed=n + 7)
    rng = np.random.default_rng(p)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    prob = R._OnlyUnitDiagProblem(C, n, p)
    h = lib.Handle.onlyunitdiag(C)
    h.set_option("dense_sym", 2); h.set_option("dense_sym_rt", shape)
    h.set_point(Y)
    f, G, H = h.cost(), h.rgrad(), h.hessvec(U)
    assert abs(f - prob.cost(Y)) <= 1e-12 * max(1.0, abs(prob.cost(Y)))
    assert _relerr(G, prob.grad(Y)) < 1e-12
    assert _relerr(H, prob.hess(Y, U)) < 1e-12
    assert np.array_equal(H, h.hessvec(U))
    h.set_option("dense_sym", 0)
    h.set_point(Y)
    assert _relerr(H, h.hessvec(U)) < 1e-13
    h.close()


def test_symmetric_contraction_short_slices(lib):
    """Work items of two steps (dense_sym_len): many D slabs per row block, slices that end inside the block on the diagonal."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    n, p = 700, 24
    C = problems.dense_unitdiag_cost(n, seed=3)
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    prob = R._OnlyUnitDiagProblem(C, n, p)
    prob.cost(Y)                                              # the closures share eG (ManiSDP_onlyunitdiag.m:118-119)
    for shape in (1, 2, 3):
        for db in (1, 2):                                      # two barriers per step / one (double-buffered reduction)
            for length in (2, 0):
                h = lib.Handle.onlyunitdiag(C)
                h.set_option("dense_sym", 2); h.set_option("dense_sym_rt", shape); h.set_option("dense_sym_len", length)
                h.set_option("dense_sym_db", db)
                h.set_point(Y)
                H = h.hessvec(U)
                assert _relerr(H, prob.hess(Y, U)) < 1e-12
                assert _relerr(h.rgrad(), prob.grad(Y)) < 1e-12
                assert np.array_equal(H, h.hessvec(U))
                h.close()


def test_asymmetric_dense_cost_keeps_the_full_kernel(lib):
    """A dense C that is not symmetric entry by entry must not take the upper-triangle route (the reference's U*C uses all of C)."""
    n, p = 300, 8
    rng = np.random.default_rng(0)
    C = rng.standard_normal((n, n))
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    out = []
    for sym in (2, 0):
        h = lib.Handle.onlyunitdiag(C)
        h.set_option("dense_sym", sym)
        h.set_point(Y)
        out.append(h.hessvec(U))
        h.close()
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("shape", [0, 1, 2, 3])
def test_trustregions_through_the_symmetric_contraction(lib, shape):
    """A whole trustregions() call (graph-replayed tCG trips, retractions, cost evaluations) and a whole ManiSDP_onlyunitdiag solve with
    every dense product on the upper-triangle route (forced below its size threshold) against the full kernel: same Hess-vec counts
    and accept / reject sequence, costs and end points equal to rounding; the solve reaches the SDPLIB value of mcp250-1."""
    import json
    from conftest import golden_path
    from manisdp_matlab_amd import problems, solvers
    n, p = 1500, 24
    C = problems.dense_unitdiag_cost(n, seed=11)
    rng = np.random.default_rng(2)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    out = []
    for sym in (2, 0):
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("dense_sym", sym); h.set_option("dense_sym_rt", shape)
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=8, maxinner=30, tolgradnorm=1e-9))
        out.append(((st.hessvecs, st.accepted, st.rejected, st.iters), st.cost, h.get_point()))
        h.close()
    assert out[0][0] == out[1][0]
    assert abs(out[0][1] - out[1][1]) <= 1e-11 * abs(out[1][1])
    assert _relerr(out[0][2], out[1][2]) < 1e-8
    if shape == 0:
        known = json.load(open(golden_path("known_answers.json")))
        At, b, c, K = problems.from_sdpa(golden_path("mcp250-1.dat-s.gz"))
        Cm = c.toarray().reshape(K["s"], K["s"], order="F")
        Yf, obj, data = solvers.ManiSDP_onlyunitdiag(Cm, {"device_options": {"dense_sym": 2}}, verbose=False)
        assert data["dinf"] < 1e-8 and abs(-obj - known["mcp250-1"]) < 1e-6 * abs(known["mcp250-1"])


@pytest.mark.parametrize("n", [8192, 20000])
def test_dense_operators_at_the_benchmarked_size(lib, n):
    """bench.py quotes k_dense_sym at n = 20000 and the route switches on by default at n >= 8192, but no
    test compared the one-GPU dense path with the oracle beyond n = 3000 (79 row blocks and a multi-round slice plan at 20000: a
    different plan).  Synthetic dense C of bench.py (generated on the device; problems.SyntheticDenseC restates the generator on
    the host row by row), p = 16 / 32 on the symmetric route (the default here) and p = 64 on the full kernel: Hess-vec and
    gradient against the oracle's closures (ManiSDP_onlyunitdiag.m:117-130) on 96 sampled rows incl. the row-block edges, the cost
    against the oracle's sum over ALL rows, two runs bit for bit, and the symmetric route against the full kernel on every row."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    seed = 0
    S = problems.SyntheticDenseC(n, seed)
    rng = np.random.default_rng(n)
    ps = (16, 32, 64)
    Ys, Us = {}, {}
    for p in ps:
        Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        Ys[p], Us[p] = Y, rng.standard_normal((n, p))
    edge = [0, 1, 15, 16, 127, 128, 255, 256, 257, 511, 512, n // 2 - 1, n // 2, n - 257, n - 256, n - 17, n - 16, n - 2, n - 1]
    rows = np.unique(np.concatenate([np.array(edge), rng.choice(n, 96 - len(edge), replace=False)]))
    # oracle: eG over all rows (for the cost), in chunks of rows of the generator; gradient / Hess-vec on the sampled rows
    f_ref = {p: 0.0 for p in ps}
    for r0 in range(0, n, 512):
        blk = np.arange(r0, min(n, r0 + 512))
        Cb = S.rows(blk)
        for p in ps:
            f_ref[p] += 0.5 * float(np.sum(R.onlyunitdiag_rows(Cb, blk, Ys[p])[0]))
    Cr = S.rows(rows)
    h = lib.Handle.dense_synthetic(n, seed, pcap=64)
    for p in ps:
        Y, U = Ys[p], Us[p]
        _, G_ref, H_ref = R.onlyunitdiag_rows(Cr, rows, Y, U)
        h.set_option("dense_sym", 1)                               # default: symmetric route from 8192 rows on, p <= 32
        h.set_point(Y)
        f, G, H = h.cost(), h.rgrad(), h.hessvec(U)
        assert abs(f - f_ref[p]) <= 1e-12 * max(1.0, abs(f_ref[p])), (p, f, f_ref[p])
        assert _relerr(G[rows], G_ref) < 1e-12, p
        assert _relerr(H[rows], H_ref) < 1e-12, p
        assert np.array_equal(H, h.hessvec(U)), p                  # no atomics, fixed summation orders
        assert np.array_equal(G, h.rgrad()), p
        if p <= 32:
            h.set_option("dense_sym", 0)                           # the full kernel on the same data: every row
            h.set_point(Y)
            assert abs(h.cost() - f) <= 1e-13 * max(1.0, abs(f))
            assert _relerr(h.rgrad(), G) < 1e-13 and _relerr(h.hessvec(U), H) < 1e-13, p
    h.close()
