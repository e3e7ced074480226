"""Row-sharded ranks in DIFFERENT PROCESSES (msdp_comm_init_ipc): the N-GPU form of the cross-rank persistent tCG as far as one GPU can
validate it -- N fresh processes share the GPU, the slot regions and the exchange buffer of the tCG live in one fine-grained device
block that rank 0 exports with hipIpcGetMemHandle and the others map with hipIpcOpenMemHandle (the mapping goes over peer access when
the ranks own different devices: the same code path), every process launches its own workgroups, and the collectives outside the tCG
go through staging slabs of the same block.  Reference: one unsharded handle (tCG.m:160-289, trustregions.m:441-767)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_counter = [0]


def _run_ranks(N, rows, cols, p, tmp_path, devices=None, two_level=False):
    """Runs the group; more than four processes on ONE device get a second try when a member's launch met a synchronisation time-out.
    Eight processes own eight sets of hardware queues on one GPU and the scheduler time-slices them when they are oversubscribed: a
    member's launch can stay unmapped for seconds while the others spin for it (seen once in six to nine runs of the 8-process case,
    never with up to four processes) -- a property of the stand-in, not of the N-GPU form, where every member owns its device."""
    tries = 2 if (N > 4 and devices is None) else 1
    for attempt in range(tries):
        sub = tmp_path / ("try%d" % attempt)
        sub.mkdir()
        try:
            return _run_ranks_once(N, rows, cols, p, sub, devices, two_level)
        except AssertionError as e:
            # (the member whose launch starved reports the time-out; the others may see the group it broke, or miss it in a barrier)
            if attempt + 1 < tries and any(m in str(e) for m in ("a grid synchronisation timed out", "group broken", "did not reach")):
                print("8 processes on one device: a member's launch was not scheduled in time -- one more try")
                continue
            raise


def _run_ranks_once(N, rows, cols, p, tmp_path, devices=None, two_level=False):
    _counter[0] += 1
    name = "/msdp_test_%d_%d" % (os.getpid(), _counter[0])
    procs, outs = [], []
    env = dict(os.environ)
    env.setdefault("MSDP_LOCAL_BARRIER_TIMEOUT", "180")     # (a fresh box pages the libraries in: eight processes starting at once can be a minute apart)
    if two_level:
        env["MSDP_TEST_XR_TWOLEVEL"] = "1"
    if N > 4 and devices is None:
        env["MSDP_TEST_LIGHT"] = "1"                # (eight processes share one device: see the worker)
    for r in range(N):
        out = str(tmp_path / ("rank%d.npz" % r))
        outs.append(out)
        cmd = [sys.executable, os.path.join(ROOT, "tests", "ipc_rank_worker.py"), str(r), str(N), name, str(rows), str(cols), str(p), out]
        if devices is not None:
            cmd.append(str(devices[r]))
        # (output to a file per worker: a pipe nobody drains while the members wait for each other in a barrier would stall all of them)
        procs.append(subprocess.Popen(cmd, env=env, stdout=open(str(tmp_path / ("rank%d.log" % r)), "w"), stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for pr in procs:
            try:
                pr.wait(timeout=600)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
        logs = [open(str(tmp_path / ("rank%d.log" % r))).read() for r in range(N)]
    finally:
        # the worker ends with a provoked time-out and skips h.close(): rank 0 never unlinks the group's segment -- the harness does
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    return [np.load(o) for o in outs]


def _reference(rows, cols, p):
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(rows, cols, seed=7)
    n = C.shape[0]
    rng = np.random.default_rng(3)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y0)
    f0, G0 = h.cost(), h.rgrad()
    st = h.rtr(_lib.default_opts(maxiter=10, maxinner=60, tolgradnorm=1e-9))
    Y1 = h.get_point()
    h.close()
    return f0, G0, st, Y1


def _check(res, ref):
    f0, G0, st, Y1 = ref
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for q in res:
        r0, r1 = q["rows"]
        assert abs(float(q["f0"]) - f0) <= 1e-12 * abs(f0)
        assert rel(q["G0"][r0:r1], G0[r0:r1]) < 1e-12
        assert int(q["path"]) == 2                                 # the cross-rank persistent kernel ran
        assert tuple(int(v) for v in q["stats"]) == (st.hessvecs, st.accepted, st.rejected, st.iters, st.last_stop_inner)
        assert abs(float(q["cost"]) - st.cost) <= 1e-10 * abs(st.cost)
        assert rel(q["Y"], Y1) < 1e-8
        assert np.array_equal(q["Y"], res[0]["Y"])                 # every member ends with the same bits
        # zero collectives per trip AND per TR iteration (cross-rank TR tail): a trustregions() call issues the set-up exchanges only,
        # whether its tCGs make 7 trips or 60 and however many iterations it runs; with xtail = 0 the iterations pay their collectives
        # and reach the same point
        assert int(q["hv7"]) < st.hessvecs
        assert int(q["calls"]) <= 6 and int(q["calls"]) == int(q["calls7"]), (int(q["calls"]), int(q["calls7"]))
        assert int(q["calls_coll"]) >= 3 * int(q["iters"])
        assert tuple(int(v) for v in q["stats_coll"]) == tuple(int(v) for v in q["stats"])
        assert rel(q["Ycoll"], q["Y"]) < 1e-9
        assert "timed out" in str(q["err"]) or "did not reach" in str(q["err"]) or "group broken" in str(q["err"]), str(q["err"])
    print("process ranks: trip %.2f us; trustregions() %.2f us per Hess-vec with the cross-rank tail, %.2f with per-iteration collectives"
          % (max(float(q["trip_us"]) for q in res), max(float(q["rtr_us_per_hv"]) for q in res), max(float(q["rtr_us_per_hv_coll"]) for q in res)))


@pytest.mark.parametrize("N,shape,p", [(2, (200, 200), 32), (2, (61, 50), 12), (4, (200, 200), 16)])
def test_process_ranks_on_one_gpu_run_one_persistent_tcg(tmp_path, N, shape, p):
    """N processes on ONE GPU, 256 / N workgroups each, separate launches: same counts / stop codes / end point as one handle, zero
    collectives per trip, a launch that waits for workgroups that do not exist ends in MSDP_ECOMM instead of hanging."""
    res = _run_ranks(N, shape[0], shape[1], p, tmp_path)
    _check(res, _reference(shape[0], shape[1], p))
    assert max(float(q["trip_us"]) for q in res) < 40.0


@pytest.mark.parametrize("N,shape,p", [(8, (100, 200), 32), (8, (100, 200), 16), (2, (200, 200), 32), (3, (61, 50), 12)])
def test_process_ranks_with_two_level_reductions(tmp_path, N, shape, p):
    """Round 6: the N-GPU form of the grid reductions (msdp_psync.h psync2) -- every member reduces over ITS OWN workgroups in its own
    block, its leader pushes the member's sums into every member's block, everybody polls its own -- as far as one GPU can run it: 8
    PROCESSES x 2 500 rows with 32 workgroups each (more than four members take this form by themselves), and the shapes of the flat
    protocol with the option xr_twolevel.  Same checks: counts / stop codes / end point of one unsharded handle, the same bits on every
    member, zero collectives per trip and per TR iteration, a launch that waits for workgroups that do not exist ends in MSDP_ECOMM."""
    res = _run_ranks(N, shape[0], shape[1], p, tmp_path, two_level=(N <= 4))
    _check(res, _reference(shape[0], shape[1], p))
    # (no bound on the time with eight processes on ONE device: their eight hardware queues are time-sliced by the scheduler, and a
    # launch whose peers are descheduled spins until they come back -- milliseconds per call, an artefact of this test bed; on eight
    # devices every member has its own)
    if N <= 4:
        assert max(float(q["trip_us"]) for q in res) < 60.0


@pytest.mark.parametrize("N", [2, 4, 8])
def test_process_ranks_on_separate_gpus(tmp_path, N):
    """The identical code path with one rank per GPU: every block and buffer is mapped over peer access, the reductions are two-level,
    the pushed rows and member sums cross xGMI with system scope (skipped where fewer than N GPUs are visible)."""
    from manisdp_matlab_amd import _lib
    _lib.load()
    if _lib.device_count() < N:
        pytest.skip("needs %d GPUs" % N)
    res = _run_ranks(N, 100 * N, 200, 32, tmp_path, devices=list(range(N)))
    _check(res, _reference(100 * N, 200, 32))
