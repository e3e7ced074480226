"""GPU parity tests for the onlyunitdiag hot path: every call goes through the
C-ABI (ctypes) and is compared with the oracle on the same seeded inputs.
Tolerances: operator outputs 1e-12 relative (fp64, different summation order);
solver-level optimum / KKT residues 1e-6 relative as BASELINE.json's north_star states."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_path

pytestmark = pytest.mark.gpu


def _rand_point(n, p, seed):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, p))
    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    return Y, U


def _relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def lib():
    from manisdp_matlab_amd import _lib
    _lib.load()
    return _lib


@pytest.mark.parametrize("p", [1, 2, 3, 5, 8, 16, 27, 32, 33, 64, 100, 130])
def test_operators_match_oracle_G1(lib, p):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n = C.shape[0]
    Y, U = _rand_point(n, p, seed=p)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    prob = R._OnlyUnitDiagProblem(C, n, p)
    f_ref = prob.cost(Y)
    G_ref = prob.grad(Y)
    assert abs(h.cost() - f_ref) <= 1e-12 * max(1.0, abs(f_ref))
    assert _relerr(h.rgrad(), G_ref) < 1e-12
    H_ref = prob.hess(Y, U)
    assert _relerr(h.hessvec(U), H_ref) < 1e-12
    assert _relerr(h.proj(U), prob.M.proj(Y, U)) < 1e-13
    assert _relerr(h.retr(U), prob.M.retr(Y, U)) < 1e-13
    assert _relerr(h.get_z(), np.sum((C @ Y) * Y, axis=1)) < 1e-12
    assert _relerr(h.get_point(), Y) == 0.0
    h.close()


def test_operators_G81_p32(lib):
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    C = problems.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    n = C.shape[0]
    Y, U = _rand_point(n, 32, seed=0)
    U -= Y * np.sum(Y * U, axis=1, keepdims=True)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    assert _relerr(h.hessvec(U), R.hessvec_onlyunitdiag(C, Y, U)) < 1e-12
    # size-independent properties: linearity and symmetry of the Riemannian Hessian on the tangent space
    V = np.random.default_rng(5).standard_normal((n, 32))
    V -= Y * np.sum(Y * V, axis=1, keepdims=True)
    HU, HV = h.hessvec(U), h.hessvec(V)
    assert _relerr(h.hessvec(2.0 * U - 3.0 * V), 2.0 * HU - 3.0 * HV) < 1e-12
    assert abs(np.sum(V * HU) - np.sum(U * HV)) < 1e-9 * abs(np.sum(V * HU))
    h.close()


def test_rtr_matches_oracle_small(lib):
    """One trustregions() call from the same start point: same cost, same gradient norm
    order, comparable Hess-vec count (iterate-level identity is not expected: summation order)."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    C = problems.toroidal_grid_maxcut(20, 30, seed=1)
    n, p = C.shape[0], 8
    Y, _ = _rand_point(n, p, seed=2)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    st = h.rtr(lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 40, 100, 1e-8)
    assert abs(st.cost - f_ref) < 1e-6 * max(1.0, abs(f_ref))
    Yg = h.get_point()
    assert np.allclose(np.linalg.norm(Yg, axis=1), 1.0, atol=1e-14)
    assert abs(h.cost() - st.cost) < 1e-10 * max(1.0, abs(st.cost))
    # the first TR iterations are deterministic enough to agree exactly in count
    assert st.iters > 0 and st.hessvecs > 0
    h.close()


@pytest.mark.parametrize("p", [4, 12, 20])
def test_csr_rows_entry_parallel_lanes_match_the_row_per_lane_group_form(lib, p):
    """G1 (800 rows of ~48 entries: CSR rows in the persistent tCG).  Round 6: the lane groups of a wave share a row and split its
    entries (option persist_ep, default on where the rows leave lanes free) -- in the two-reduction trip and, new, in the one-reduction
    trip (persist_form() == 2 on G1).  Same step as with one lane group per row (persist_ep = 0), and as the oracle (tCG.m:160-289 inside
    trustregions.m:441-767): counts and stop code equal, cost to 1e-11."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=11)
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 3, 20, 1e-8)
    h = lib.Handle.onlyunitdiag(C, pcap=p)
    got = []
    for ep in (1, 0):
        h.set_option("persist_ep", ep)
        for pipe in (1, 0):                                          # one / two grid reductions per trip (msdp_pipe.h needs the shared rows)
            h.set_option("persist_pipe", pipe)
            for fused in (1, 0):
                h.set_option("fused_rtr", fused)
                h.set_point(Y)
                assert h.tcg_path() == 1
                assert h.persist_form() == (2 if ep and pipe else 0), (ep, pipe)
                st = h.rtr(lib.default_opts(maxiter=3, maxinner=20, tolgradnorm=1e-8))
                assert (st.iters, st.hessvecs, st.last_stop_inner) == (info.iters, info.hessvecs, info.stop_inner[-1]), (ep, pipe, fused)
                assert abs(st.cost - f_ref) < 1e-11 * max(1.0, abs(f_ref)), (ep, pipe, fused)
                got.append(h.get_point())
    for Yg in got[1:]:
        assert np.abs(Yg - got[0]).max() < 1e-9
    h.close()


def test_rtr_first_iteration_trace(lib):
    """With maxiter = 1 the solve is a single tCG: Hess-vec count and cost must agree with the
    oracle to rounding (no accumulated divergence yet)."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n, p = C.shape[0], 4
    Y, _ = _rand_point(n, p, seed=7)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    for maxinner in (1, 3, 10):
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
        prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        assert st.hessvecs == info.hessvecs
        assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-11 * max(1.0, abs(f_ref))
        assert abs(st.gradnorm - info.gradnorm) < 1e-9 * max(1.0, info.gradnorm)
    h.close()


@pytest.mark.parametrize("name,key", [("mcp100", "mcp100"), ("mcp124-1", "mcp124-1"), ("mcp250-1", "mcp250-1")])
def test_known_answers_mcp(lib, name, key):
    """SDPLIB optimal values shipped with the reference (data/sdplib/README:76-88)."""
    from manisdp_matlab_amd import problems, solvers
    known = json.load(open(golden_path("known_answers.json")))
    At, b, c, K = problems.from_sdpa(golden_path(name + ".dat-s.gz"))
    n = K["s"]
    C = sp.csr_matrix(c.toarray().reshape(n, n, order="F"))
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {}, verbose=False)
    assert data["dinf"] < 1e-8 and data["status"] == 0
    assert abs(-obj - known[key]) < 1e-6 * abs(known[key])      # 7 printed digits


def test_solver_G1_matches_oracle(lib):
    from manisdp_matlab_amd import problems, solvers
    from oracle import manisdp_ref as R
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    Y0, _ = _rand_point(C.shape[0], 2, seed=0)
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"Y0": Y0}, verbose=False)
    Yr, objr, datar = R.ManiSDP_onlyunitdiag(C, {"Y0": Y0}, q1="correct")
    assert data["dinf"] < 1e-8 and datar["dinf"] < 1e-8
    assert abs(obj - objr) < 1e-6 * abs(objr)
    assert abs(obj - (-12083.19765455)) < 1e-6 * 12083.2       # SURVEY.md probe value
    Y, obj11, d11 = solvers.ManiSDP_onlyunitdiag(problems.maxcut_cost_matrix(golden_path("G11.txt.gz")), {}, verbose=False)
    known = json.load(open(golden_path("known_answers.json")))
    assert abs(-obj11 - known["maxG11"]) < 1e-6 * known["maxG11"]


def _ring_lattice_cost(n, k, seed):
    """Symmetric sparse C with 2k off-diagonal entries + the diagonal per row (ELL width 2k+1)."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for d in range(1, k + 1):
        w = rng.standard_normal(n)
        i = np.arange(n)
        j = (i + d) % n
        rows += [i, j]; cols += [j, i]; vals += [w, w]
    rows.append(np.arange(n)); cols.append(np.arange(n)); vals.append(rng.standard_normal(n))
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


@pytest.mark.parametrize("shape,p,k", [((20, 30), 4, 0), ((20, 30), 16, 0), ((33, 37), 20, 0), ((20, 30), 32, 0),
                                      ((25, 40), 40, 0), ((20, 30), 64, 0), ((1, 997), 12, 3), ((1, 2500), 32, 2), ((1, 1201), 24, 3), ((1, 700), 10, 4)])
def test_persistent_tcg_matches_oracle(lib, shape, p, k):
    """The single-launch persistent tCG kernel (msdp_persist.hip) against the oracle's tCG: with maxiter = 1
    the solve is ONE tCG, so the Hess-vec count, the stop reason and the cost after the step must agree to
    rounding for every inner-iteration cap.  Covers LPR = 8/16/32, both stored ELL widths (5 and 8),
    ragged row chunks and pad lanes (p = 12, 20, 40)."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=3) if k == 0 else _ring_lattice_cost(shape[1], k, seed=4)
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=11)
    h = lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    assert h.tcg_path() == 1, "persistent kernel not selected"
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    # both trip forms of the kernel (round 5): ONE grid reduction per trip (msdp_pipe.h: p <= 32; the default there) and the
    # two-reduction trip (p > 32, and persist_pipe = 0)
    one_reduction = p <= 32                                  # (rows of <= 5 entries, of 6..8 in the stored width 8, and -- round 6 -- CSR rows with entry-parallel lanes: nine entries at n = 700)
    refs = {}
    # ... each as per-iteration launches (tCG kernel + TR tail kernel) and, at p <= 32, with the whole trustregions() loop in one launch
    for pipe, fused in ((1, 1), (1, 0), (0, 1)):
        h.set_option("persist_pipe", pipe)
        h.set_option("fused_rtr", fused)
        h.set_point(Y)
        assert h.persist_form() == (2 if pipe and one_reduction else 0)
        for maxinner in (1, 2, 7, 100):
            h.set_point(Y)
            st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
            if maxinner not in refs:
                refs[maxinner] = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
            _, f_ref, info = refs[maxinner]
            assert st.hessvecs == info.hessvecs
            assert st.last_stop_inner == info.stop_inner[-1]
            assert abs(st.cost - f_ref) < 1e-11 * max(1.0, abs(f_ref))
            assert abs(st.gradnorm - info.gradnorm) < 1e-8 * max(1.0, info.gradnorm)
        # a full solve lands on the same optimum as the chunked path's oracle
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        if "full" not in refs:
            refs["full"] = manopt_rtr.trustregions(prob, Y.copy(), 40, 100, 1e-8)
        _, f_ref, info = refs["full"]
        assert abs(st.cost - f_ref) < 1e-6 * max(1.0, abs(f_ref))
        assert np.allclose(np.linalg.norm(h.get_point(), axis=1), 1.0, atol=1e-14)
    h.close()


def test_persistent_path_selection(lib):
    """CSR rows of any length run in the persistent kernel too (G1: up to ~50 nonzeros per row); p > 64 keeps the chunked
    path, and so do more rows per workgroup than the largest row-slot instance of the width holds (p = 33..64: 128,
    p <= 32: 256 -- on 256 CUs n = 40000 at p = 32 is persistent, n = 62500 at p = 40 is not)."""
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    Y, _ = _rand_point(C.shape[0], 8, seed=1)
    h = lib.Handle.onlyunitdiag(C)
    h.set_point(Y)
    assert h.tcg_path() == 1
    h.close()
    C = problems.toroidal_grid_maxcut(20, 30, seed=3)
    Y, _ = _rand_point(C.shape[0], 80, seed=1)
    h = lib.Handle.onlyunitdiag(C, pcap=80)
    h.set_point(Y)
    assert h.tcg_path() == 0
    h.close()
    C = problems.toroidal_grid_maxcut(200, 200, seed=3)
    Y, _ = _rand_point(C.shape[0], 32, seed=1)
    h = lib.Handle.onlyunitdiag(C, pcap=32)
    h.set_point(Y)
    assert h.tcg_path() == 1
    h.close()
    C = problems.toroidal_grid_maxcut(250, 250, seed=3)
    Y, _ = _rand_point(C.shape[0], 40, seed=1)
    h = lib.Handle.onlyunitdiag(C, pcap=40)
    h.set_point(Y)
    assert h.tcg_path() == 0
    h.close()


@pytest.mark.parametrize("p", [3, 16, 24, 50])
def test_persistent_tcg_csr_rows_G1(lib, p):
    """Persistent kernel, CSR mode (rows longer than the ELL limit), against the oracle's single tCG."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=5)
    h = lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    assert h.tcg_path() == 1
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    for maxinner in (1, 6, 100):
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        assert st.hessvecs == info.hessvecs
        assert st.last_stop_inner == info.stop_inner[-1]
        assert abs(st.cost - f_ref) < 1e-11 * max(1.0, abs(f_ref))
    h.close()


def test_fused_rtr_kernel_matches_per_iteration_path(lib):
    """The whole trustregions() loop in one launch (option fused_rtr, the default for p <= 32) takes the same decisions
    as one persistent launch per TR iteration: same iteration / Hess-vec / accept counts and the same cost to rounding."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(30, 40, seed=2)
    n, p = C.shape[0], 10
    Y, _ = _rand_point(n, p, seed=1)
    out = []
    for fused in (0, 1):
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("fused_rtr", fused)
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=6, maxinner=40, tolgradnorm=1e-9))
        out.append((st.iters, st.hessvecs, st.accepted, st.rejected, st.cost, st.gradnorm, h.cost()))
        h.close()
    a, b = out
    assert a[:4] == b[:4]
    assert abs(a[4] - b[4]) < 1e-11 * max(1.0, abs(a[4])) and abs(a[5] - b[5]) < 1e-8 * max(1.0, a[5])
    assert abs(b[6] - b[4]) < 1e-10 * max(1.0, abs(b[4]))       # the resident point IS the accepted one


@pytest.mark.parametrize("p", [16, 32, 40])
def test_persistent_tcg_keeps_heta_equal_to_hess_eta_on_G81(lib, p):
    """the persistent kernel assembles C*mdelta_new by linearity from the gathered residual rows and
    beta*(C*mdelta_old) (msdp_persist.hip, TWOSYNC) while mdelta_new is re-projected (tCG.m:273,283), and the assembled
    product used to drift: tCG's invariant Heta = Hess(eta) (tCG.m:192-220: both are updated with the same alpha from mdelta
    and Hess*mdelta -- every Hmdelta the kernel uses enters Heta) was off by 3e-9 after 100 trips on G81 where the direct
    products of the chunked path stay at 1e-13.  Round 3: the neighbours gather tangent(r_new) and every 32nd trip exchanges
    the direction itself (option persist_refresh).  G81 (n = 20000, ill-conditioned) near a stationary point, ONE tCG of 50
    and of 100 trips (the reference's TR_maxinner): |Heta - Hess(eta)| / |Heta| for the persistent kernel and for both chunked
    trips (every product a direct gather; the cancellation inside Heta = sum of alpha*Hmdelta sets their 1e-13)."""
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=0)
    devs = {}
    # "persistent": the kernel's default trip -- ONE grid reduction per trip at p <= 32 (msdp_pipe.h: two products by recurrence, afresh
    # from direct gathers every 16th trip), two reductions at p = 40; "persistent-two": the two-reduction trip everywhere
    for name, persist, trip2 in (("persistent", 1, 1), ("persistent-two", 1, 1), ("two-launch", 0, 1), ("three-launch", 0, 0), ("sharded", 0, 0), ("linear", 0, 0)):
        if name == "persistent-two" and p > 32:
            continue
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist_pipe", 0 if name == "persistent-two" else 1)
        if name == "sharded":
            # the row-sharded trip with one all-reduce (msdp_trip1.hip) assembles its products by the same linearity, with the
            # same refresh schedule; a communicator of one in-process member runs exactly that code
            h.comm_init_local(1, 0, 7700 + p)
        h.set_option("persist", persist)
        # "linear": the same trip on one rank without communicator (the chunked path's default from 2^21 vector entries on),
        # replayed from hipGraphs with the refresh launches appended behind every fourth replay
        h.set_option("trip1", 2 if name == "linear" else (1 if name == "sharded" else 0))
        h.set_option("trip2", 2 * trip2)
        h.set_option("fused_rtr", 0)                       # the step is handed over through global memory
        h.set_point(Y)
        # towards a stationary point, until a tCG from there runs its whole budget (away from one it leaves through negative
        # curvature after a dozen trips)
        o = lib.default_opts(maxiter=1, maxinner=100, tolgradnorm=1e-14)
        o.Delta0 = 1e3; o.Delta_bar = 1e6                      # no boundary exit
        for _ in range(12):
            h.rtr(lib.default_opts(maxiter=20, maxinner=100, tolgradnorm=1e-8))
            Yc = h.get_point()
            full = h.rtr(o).hessvecs == 100
            h.set_point(Yc)
            if full:
                break
        assert h.tcg_path() == persist
        if persist:
            assert h.persist_form() == (2 if name == "persistent" and p <= 32 else 0)
        for trips in (50, 100):
            h.set_point(Yc)
            o.maxinner = trips
            st = h.rtr(o)
            eta, heta = h.debug_get_tcg_step()
            h.set_point(Yc)
            h.cost()
            He = h.hessvec(eta)
            dev = np.linalg.norm(heta - He) / np.linalg.norm(heta)
            devs[(name, trips)] = (dev, st.hessvecs, st.last_stop_inner)
        h.close()
    for (name, trips), (dev, hv, stop) in devs.items():
        assert hv == trips, (name, trips, hv, stop)                 # the whole budget, not an early exit
        assert dev <= (2e-11 if name in ("persistent", "sharded", "linear") else 2e-12), (name, trips, dev, hv, stop)


@pytest.mark.parametrize("p", [8, 16, 32])
def test_one_reduction_trip_solves_G81_like_the_two_reduction_trip(lib, p):
    """Round 5 (msdp_pipe.h): the persistent tCG with ONE grid reduction per trip -- the values of tCG.m:227-241 expanded in the step
    length, the rows of H*mdelta published before the reduction, C*tangent(r) and C*mdelta by linearity -- against the two-reduction
    trip on whole trustregions() calls on G81 (40 iterations, inner cap 100: negative-curvature / boundary exits, model and residual
    stops all occur): same iteration, Hess-vec, accept / reject counts and last stop code, cost and gradient norm to rounding, the
    end point within the tolerance of the north star; two runs of the one-reduction trip bit for bit; every refresh interval."""
    from manisdp_matlab_amd import problems
    C = problems.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=0)
    h = lib.Handle.onlyunitdiag(C, pcap=p)
    opts = lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    out = {}
    for name, pipe, refresh, fused in (("two", 0, 16, 0), ("one", 1, 16, 0), ("one again", 1, 16, 0), ("one/1", 1, 1, 0), ("one/2", 1, 2, 0), ("one/4", 1, 4, 0), ("one/64", 1, 64, 0),
                                       ("one fused", 1, 16, 1), ("one fused again", 1, 16, 1), ("one all rows through the buffer", 1, 16, 0)):
        h.set_option("persist_pipe", pipe)
        h.set_option("pipe_refresh", refresh)
        h.set_option("fused_rtr", fused)                       # 1: the whole trustregions() loop in the one launch (k_tcg_pipe_obl<.., FUSE>)
        h.set_option("pipe_local", 0 if "buffer" in name else 1)
        h.set_point(Y)
        assert h.tcg_path() == 1 and h.persist_form() == (2 if pipe else 0)
        st = h.rtr(opts)
        out[name] = (st, h.get_point())
    ref, Yref = out["two"]
    # ("one/1": a refresh interval of 1 is raised to 2 by msdp_set_option -- with 1 a workgroup would publish the refresh rows of trip
    # j + 1 into the regions a slower one still gathers those of trip j from)
    assert np.array_equal(out["one/1"][1], out["one/2"][1])
    for name in ("one", "one/1", "one/2", "one/4", "one/64", "one fused", "one all rows through the buffer"):
        st, Yo = out[name]
        assert (st.iters, st.hessvecs, st.accepted, st.rejected, st.last_stop_inner) == (ref.iters, ref.hessvecs, ref.accepted, ref.rejected, ref.last_stop_inner), name
        assert abs(st.cost - ref.cost) <= 1e-11 * abs(ref.cost), (name, st.cost, ref.cost)
        assert abs(st.gradnorm - ref.gradnorm) <= 1e-6 * ref.gradnorm, (name, st.gradnorm, ref.gradnorm)
        assert np.abs(Yo - Yref).max() <= 1e-6, name
        assert np.allclose(np.linalg.norm(Yo, axis=1), 1.0, atol=1e-14)
    assert np.array_equal(out["one"][1], out["one again"][1]) and out["one"][0].cost == out["one again"][0].cost
    assert np.array_equal(out["one fused"][1], out["one fused again"][1]) and out["one fused"][0].cost == out["one fused again"][0].cost
    h.close()


@pytest.mark.parametrize("shape,p,k", [((20, 30), 3, 0), ((20, 30), 16, 0), ((33, 37), 20, 0), ((25, 40), 40, 0), ((20, 30), 80, 0),
                                      ((12, 25), 150, 0), ((1, 997), 12, 3), ((0, 0), 24, -1)])
def test_two_launch_trip_matches_oracle_and_three_launch_trip(lib, shape, p, k):
    """The two-launch tCG trip of the chunked path (msdp_trip2.hip: Heta implied as r - grad, the new direction recomputed for
    the gathered rows) against the oracle's tCG and against the three-launch trip it replaces: with maxiter = 1 a solve is ONE
    tCG, so Hess-vec count, stop code, cost and the step itself (eta, Heta) must agree for every inner-iteration cap -- also
    the caps that end exactly on a chunk boundary (8, 16).  ELL rows, CSR rows (ring lattice with 7 entries, G1 with ~48), pad
    lanes (p = 3, 12, 20) and multi-chunk rows (p = 150)."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R, manopt_rtr
    if k < 0:
        C = problems.maxcut_cost_matrix(golden_path("G1.txt.gz"))
    else:
        C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=3) if k == 0 else _ring_lattice_cost(shape[1], k, seed=4)
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=11)
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    hs = []
    for trip2, trip1 in ((1, 0), (0, 0), (0, 2)):
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("persist", 0)
        h.set_option("trip2", 2 * trip2)                    # 2: also below the size from which it is the default
        h.set_option("trip1", trip1)                        # 2: the linear-product trip of msdp_trip1.hip on one rank
        hs.append(h)
    for maxinner in (1, 2, 7, 8, 16, 100):
        _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 1, maxinner, 1e-8)
        steps = []
        for h in hs:
            h.set_point(Y)
            assert h.tcg_path() == 0
            st = h.rtr(lib.default_opts(maxiter=1, maxinner=maxinner, tolgradnorm=1e-8))
            assert st.hessvecs == info.hessvecs
            assert st.last_stop_inner == info.stop_inner[-1]
            assert abs(st.cost - f_ref) < 1e-11 * max(1.0, abs(f_ref))
            assert abs(st.gradnorm - info.gradnorm) < 1e-8 * max(1.0, info.gradnorm)
            steps.append(h.debug_get_tcg_step())
        (e2, h2), (e3, h3), (e1, h1) = steps
        assert np.linalg.norm(e2 - e3) <= 1e-10 * max(1.0, np.linalg.norm(e3))
        assert np.linalg.norm(h2 - h3) <= 1e-10 * max(1.0, np.linalg.norm(h3))
        assert np.linalg.norm(e1 - e3) <= 1e-10 * max(1.0, np.linalg.norm(e3))
        assert np.linalg.norm(h1 - h3) <= 1e-10 * max(1.0, np.linalg.norm(h3))
    # full solves: same decisions all the way (iterations, Hess-vecs, accepted / rejected steps), same optimum
    outs = []
    for h in hs:
        h.set_point(Y)
        st = h.rtr(lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8))
        outs.append((st.iters, st.cost, st.hessvecs))
        assert np.allclose(np.linalg.norm(h.get_point(), axis=1), 1.0, atol=1e-14)
        h.close()
    _, f_ref, info = manopt_rtr.trustregions(prob, Y.copy(), 40, 100, 1e-8)
    for it, cost, hv in outs:
        assert abs(cost - f_ref) < 1e-6 * max(1.0, abs(f_ref))


@pytest.mark.parametrize("shape,p", [((223, 227), 12), ((97, 531), 40), ((1, 4099), 6)])
def test_windowed_row_traversal_covers_every_row_once(lib, shape, p):
    """Option sweep (msdp_sweep_rows, msdp_device.h): from 2^21 vector entries on, the gather launch of the chunked tCG trip walks
    the rows window by window -- the co-resident workgroups of an XCD take consecutive 64-row blocks -- instead of chunk by chunk.
    Forced here at sizes whose row counts are no multiple of anything (50 621 = 223 x 227, 51 507, 4099 rows; p = 6, 12, 40:
    4, 8 and 32 lanes per row) against the chunk traversal: the same rows get the same numbers, only the per-workgroup partial
    sums are formed over different row sets -- Hess-vec counts and stop codes equal, costs and points equal to rounding --
    with the linear-product trip (the default) and the two-launch trip."""
    from manisdp_matlab_amd import problems
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=5) if shape[0] > 1 else _ring_lattice_cost(shape[1], 3, seed=6)
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=2)
    for trip1, trip2 in ((1, 0), (0, 2)):
        out = []
        for sweep in (2, 0):
            h = lib.Handle.onlyunitdiag(C, pcap=p)
            h.set_option("persist", 0)
            h.set_option("trip1", trip1); h.set_option("trip2", trip2)
            h.set_option("sweep", sweep)
            if p == 40:
                h.set_option("grid", 512)                   # two rounds of workgroups: the two-phase form of the traversal
            h.set_point(Y)
            st = h.rtr(lib.default_opts(maxiter=6, maxinner=40, tolgradnorm=1e-9))
            out.append((st.hessvecs, st.accepted, st.rejected, st.last_stop_inner, st.cost, h.get_point()))
            h.close()
        a, b = out
        assert a[:4] == b[:4]
        assert abs(a[4] - b[4]) <= 1e-12 * abs(b[4])
        assert np.linalg.norm(a[5] - b[5]) <= 1e-9 * np.linalg.norm(b[5])


@pytest.mark.parametrize("shape,p", [((223, 227), 6), ((1, 51507), 12), ((1, 4099), 40), ((61, 67), 32)])
def test_windowed_row_traversal_of_the_standalone_operators(lib, shape, p):
    """The stand-alone cost / gradient and Hess-vec kernels (k_costgrad_*_obl, k_hess_*_obl: ManiSDP_onlyunitdiag.m:117-130) walk
    the rows with the same windowed order when option sweep applies (round 4): every row gets the same bits as with the chunk
    order -- a row's result does not depend on which workgroup computes it -- and the sums over the rows agree to rounding."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=5) if shape[0] > 1 else _ring_lattice_cost(shape[1], 3, seed=6)
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=4)
    U = np.random.default_rng(9).standard_normal((n, p))
    out = []
    for sweep in (2, 3, 0):                # 3: the software-pipelined streaming form of the Hess-vec (k_hess_ell_stream*)
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("sweep", sweep)
        if p == 40:
            h.set_option("grid", 512)
        h.set_point(Y)
        out.append((h.cost(), h.rgrad(), h.hessvec(U), h.get_z()))
        h.close()
    a, a3, b = out
    for a in (a, a3):
        assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0])
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    prob = R._OnlyUnitDiagProblem(C, n, p)
    prob.cost(Y)
    assert np.linalg.norm(a[2] - prob.hess(Y, U)) <= 1e-12 * np.linalg.norm(a[2])


@pytest.mark.parametrize("shape,p", [((200, 300), 32), ((100, 100), 16), ((141, 142), 5), ((300, 300), 40), ((60, 50), 64), ((1, 4000), 12), ((0, 3000), 8)])
def test_lds_staged_hessvec_matches_the_direct_gathers(lib, shape, p):
    """Round 5 (north_star's "LDS-staged p-wide panels"): option window = 2 routes the stand-alone Hess-vec
    (ManiSDP_onlyunitdiag.m:127-130) through k_hess_win_obl -- breadth-first patches of rows, the rows of U a patch touches loaded
    ONCE per workgroup into LDS, products formed from LDS with patch-local indices in the fma order of the direct gathers.  Every
    row must agree with k_hess_ell_obl to the last bits, on grids, on a ring lattice (rows of 7 entries) and on a random graph without
    locality (there the plan is refused and the direct kernel serves the call); sharded handles take their rows of the same plan."""
    from manisdp_matlab_amd import problems
    from oracle import manisdp_ref as R
    if shape[0] > 1:
        C = problems.toroidal_grid_maxcut(shape[0], shape[1], seed=5)
    elif shape[0] == 1:
        C = _ring_lattice_cost(shape[1], 3, seed=6)
    else:
        import scipy.sparse as sp
        n0 = shape[1]
        rng0 = np.random.default_rng(2)
        ii = rng0.integers(0, n0, 2 * n0); jj = rng0.integers(0, n0, 2 * n0)
        A = sp.coo_matrix((rng0.choice([1.0, -1.0], ii.size), (ii, jj)), shape=(n0, n0)).tocsr()
        A = A + A.T
        A.setdiag(0); A.eliminate_zeros()
        C = (sp.diags(np.asarray(abs(A).sum(axis=1)).ravel()) - A).tocsr() * -0.25
    n = C.shape[0]
    Y, _ = _rand_point(n, p, seed=4)
    U = np.random.default_rng(9).standard_normal((n, p))
    out = []
    for window in (2, 0, 3):               # 3: two workgroups per CU with one window buffer each, partial sums folded by k_win_fold
        h = lib.Handle.onlyunitdiag(C, pcap=p)
        h.set_option("window", window)
        h.set_point(Y)
        H = h.hessvec(U)
        assert np.array_equal(H, h.hessvec(U))
        out.append(H)
        if window:                         # the partial sums <U, H U> reach the tCG kernels: one trustregions() step must agree with the direct form
            st = h.rtr(lib.default_opts(maxiter=2, maxinner=6, tolgradnorm=1e-9))
            out.append((st.hessvecs, st.cost))
        h.close()
    assert out[1][0] == out[4][0] and abs(out[1][1] - out[4][1]) <= 1e-12 * abs(out[1][1])
    assert np.array_equal(out[0], out[3])
    out = [out[0], out[2]]
    # (same fma order in the row products; the row dot <Y, C*U> is summed over 8 or 16 lanes depending on the direct kernel's lane plan)
    assert np.abs(out[0] - out[1]).max() <= 1e-13 * np.abs(out[1]).max()
    prob = R._OnlyUnitDiagProblem(C, n, p)
    prob.cost(Y)
    assert np.linalg.norm(out[0] - prob.hess(Y, U)) <= 1e-12 * np.linalg.norm(out[0])
    if shape[0] > 1:
        for N in (2, 3):
            for r in range(N):
                h = lib.Handle.onlyunitdiag(C, pcap=p)
                h.debug_shard(N, r)
                h.set_option("window", 2)
                h.set_point(Y)
                r0, r1 = h.local_rows()
                h.debug_set_full_rows(Y)                           # stands in for the all-gather of the point ...
                h.rgrad()
                h.debug_set_full_rows(U)                           # ... and of the direction
                assert np.abs(h.hessvec(U)[r0:r1] - out[1][r0:r1]).max() <= 1e-13 * np.abs(out[1]).max()
                h.close()
