"""The step bench.py times, pinned to the oracle (missing 3): ONE trustregions() call of ManiSDP_onlyunitdiag on
G81 from the seed-0 start point with the reference's inner-solver options (`maxiter = 40, maxinner = 100`,
ManiSDP_onlyunitdiag.m:32-41), i.e. trustregions.m:441-767 with tCG.m:160-289 inside, for every form of the tCG the device
offers at that size -- fused / per-iteration launches, one / two grid reductions per trip, the chunked trips -- against
oracle/oracle_core.c (C restatement) and, once, oracle/manopt_rtr.py (NumPy restatement).  What must agree: the iteration count,
the Hess-vec count (997 at p = 32: bench.py's `hessvecs_per_step`), the accept / reject counts, the stop code of the last tCG --
all exactly -- the cost to 1e-10 and the gradient norm to 1e-6 relative.  The process-rank path (two processes on one GPU,
2 x 10 000 rows) is held to the same figures."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _start(n, p):
    rng = np.random.default_rng(0)                 # bench.py's Y0
    Y = rng.standard_normal((n, p))
    return Y / np.linalg.norm(Y, axis=1, keepdims=True)


def _same(st, ref, what):
    got = (st.iters, st.hessvecs, st.accepted, st.rejected, st.last_stop_inner)
    want = (ref.iters, ref.hessvecs, ref.accepted, ref.rejected, ref.last_stop_inner)
    assert got == want, (what, got, want)
    assert abs(st.cost - ref.cost) <= 1e-10 * abs(ref.cost), (what, st.cost, ref.cost)
    assert abs(st.gradnorm - ref.gradnorm) <= 1e-6 * ref.gradnorm, (what, st.gradnorm, ref.gradnorm)


# the oracle's own figures for the step (oracle_core.c, any thread count): a change of the oracle shows up here first
ORACLE_COUNTS = {8: (40, 532, 31, 9, 2), 16: (40, 1241, 32, 8, 2), 32: (40, 997, 32, 8, 2)}


@pytest.mark.parametrize("p", [8, 16, 32])
def test_benchmarked_step_matches_the_oracle_on_G81(p):
    from manisdp_matlab_amd import _lib, problems
    from oracle import core
    _lib.load()
    C = problems.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    n = C.shape[0]
    Y0 = _start(n, p)
    Yo, ref = core.rtr_onlyunitdiag(C, Y0, 40, 100, 1e-8)
    assert (ref.iters, ref.hessvecs, ref.accepted, ref.rejected, ref.last_stop_inner) == ORACLE_COUNTS[p]
    if p == 8:
        # the NumPy restatement of trustregions.m / tCG.m on the same step (the C oracle is pinned by it at small sizes:
        # tests/test_oracle_c_core.py; here at the benchmark's size)
        from oracle import manisdp_ref as R, manopt_rtr
        prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
        _, f_np, info = manopt_rtr.trustregions(prob, Y0.copy(), 40, 100, 1e-8)
        assert (info.iters, info.hessvecs, info.stop_inner[-1]) == (ref.iters, ref.hessvecs, ref.last_stop_inner)
        assert abs(f_np - ref.cost) <= 1e-10 * abs(ref.cost)
        assert abs(info.gradnorm - ref.gradnorm) <= 1e-6 * ref.gradnorm
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    # (name, persist, persist_pipe, fused_rtr, expected persist_form)
    forms = [("fused launch, one reduction per trip (bench.py's step)", 1, 1, 1, 2),
             ("per-iteration launches, one reduction per trip", 1, 1, 0, 2),
             ("fused launch, two reductions per trip", 1, 0, 1, 0),
             ("per-iteration launches, two reductions per trip", 1, 0, 0, 0),
             ("chunked trips", 0, 1, 0, -1)]
    for name, persist, pipe, fused, form in forms:
        h.set_option("persist", persist)
        h.set_option("persist_pipe", pipe)
        h.set_option("fused_rtr", fused)
        h.set_point(Y0)
        assert h.tcg_path() == persist and h.persist_form() == form, name
        st = h.rtr(opts)
        _same(st, ref, "p = %d, %s" % (p, name))
        Yg = h.get_point()
        assert np.abs(Yg - Yo).max() <= 1e-6, name                       # the accepted point itself, north_star's tolerance
        assert np.allclose(np.linalg.norm(Yg, axis=1), 1.0, atol=1e-14)
    h.close()


def test_benchmarked_step_on_process_ranks_matches_the_oracle(tmp_path):
    """Two processes on one GPU, 2 x 10 000 rows of G81 each (msdp_comm_init_ipc: the cross-rank persistent tCG and the cross-rank
    TR tail): the same step, the same figures."""
    from manisdp_matlab_amd import problems
    from oracle import core
    p = 32
    C = problems.maxcut_cost_matrix(golden_path("G81.txt.gz"))
    n = C.shape[0]
    _, ref = core.rtr_onlyunitdiag(C, _start(n, p), 40, 100, 1e-8)
    name = "/msdp_step_%d" % os.getpid()
    env = dict(os.environ)
    env.setdefault("MSDP_LOCAL_BARRIER_TIMEOUT", "180")     # (a fresh box pages the libraries in: eight processes starting at once can be a minute apart)
    procs, outs, logs = [], [], []
    try:
        for r in range(2):
            out = str(tmp_path / ("rank%d.npz" % r))
            outs.append(out)
            log = open(str(tmp_path / ("rank%d.log" % r)), "w")
            logs.append(log)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_step_worker.py"), str(r), "2", name, str(p), out],
                                          env=env, stdout=log, stderr=subprocess.STDOUT))
        for pr in procs:
            try:
                pr.wait(timeout=600)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    finally:
        for log in logs:
            log.close()
        try:
            os.unlink("/dev/shm" + name)                                  # (rank 0 unlinks it on close; this covers a crashed worker)
        except OSError:
            pass
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, "rank %d failed:\n%s" % (r, open(str(tmp_path / ("rank%d.log" % r))).read()[-3000:])
    res = [np.load(o) for o in outs]
    for q in res:
        assert int(q["path"]) == 2                                       # the cross-rank persistent kernel ran
        got = tuple(int(v) for v in q["stats"])
        assert got == (ref.iters, ref.hessvecs, ref.accepted, ref.rejected, ref.last_stop_inner), got
        assert abs(float(q["cost"]) - ref.cost) <= 1e-10 * abs(ref.cost)
        assert abs(float(q["gradnorm"]) - ref.gradnorm) <= 1e-6 * ref.gradnorm
        assert np.array_equal(q["Y"], res[0]["Y"])
