"""CPU tests of bench.py's host logic (no GPU, a fake handle): the halo leg of a multi-rank run skips its timed collectives
on EVERY rank when the set-up failed on one of them (one rank raising alone left the others in the
collectives until the watchdog), and the choice of `value` between the two row exchanges is configuration, not best-of."""
import threading

import numpy as np

import bench


class _FakeStats:
    hessvecs = 7


class _FakeHandle:
    def __init__(self, log):
        self.log = log

    def set_option(self, name, value):
        self.log.append(("opt", name, value))

    def set_point(self, Y):
        pass

    def point_snapshot(self):
        pass

    def point_restore(self):
        pass

    def rtr(self, opts):
        if self.fail_rtr_at is not None and self.log.count(("rtr",)) >= self.fail_rtr_at:
            raise RuntimeError("cross-rank persistent tCG: a grid synchronisation timed out")
        self.log.append(("rtr",))
        return _FakeStats()

    fail_rtr_at = None

    def tcg_path(self):
        return 2

    def collective_calls(self):
        return 4

    def bench_tcg_trip(self, reps):
        return 0.011

    def close(self):
        self.log.append(("close",))


def _run(N, fail_create=None, fail_join=None, leg="halo", fail_rtr=None):
    bar = threading.Barrier(N, timeout=20)
    box = [0.0] * N
    logs = [[] for _ in range(N)]
    out = [None] * N

    def one(r):
        class Lib:
            class Handle:
                @staticmethod
                def onlyunitdiag(C, pcap=32):
                    if r == fail_create:
                        raise MemoryError("no room for the halo buffers")
                    hh = _FakeHandle(logs[r])
                    if fail_rtr is not None:          # the library raises on EVERY member when a grid reduction times out
                        hh.fail_rtr_at = fail_rtr
                    return hh

        def join(h):
            if r == fail_join:
                raise RuntimeError("communicator set-up failed")

        def allmax(x):
            box[r] = x
            bar.wait()
            m = max(box)
            bar.wait()
            return m

        fn = bench.halo_leg if leg == "halo" else bench.xr_leg
        out[r] = fn(Lib, join, bar.wait, allmax, N, r, None, np.zeros((4, 2)), 2, None, 3, 1)

    ts = [threading.Thread(target=one, args=(r,)) for r in range(N)]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    assert not any(t.is_alive() for t in ts), "a rank is stuck in the halo leg"
    return out, logs


def test_halo_leg_runs_on_all_ranks():
    out, logs = _run(3)
    for o, lg in zip(out, logs):
        assert "error" not in o and o["hessvecs"] == 21 and o["value"] > 0
        assert ("opt", "halo_exchange", 1) in lg and lg.count(("rtr",)) == 4 and lg[-1] == ("close",)


def test_halo_leg_is_skipped_together_when_one_rank_fails():
    for kw in ({"fail_create": 1}, {"fail_join": 2}):
        out, logs = _run(3, **kw)
        for o, lg in zip(out, logs):
            assert "error" in o
            assert ("rtr",) not in lg                      # nobody entered the timed collectives


def test_xr_leg_runs_on_all_ranks_and_reports_the_path():
    """bench.py --gpus N, the process-group leg (cross-rank persistent kernels over HIP IPC): the same skeleton as the halo leg plus
    the figures taken behind the timed loop."""
    out, logs = _run(2, leg="xr")
    for o, lg in zip(out, logs):
        assert "error" not in o and o["hessvecs"] == 21 and o["tcg_path"] == 2
        assert abs(o["trip_us_cross_rank_persistent"] - 11.0) < 1e-9
        assert lg.count(("rtr",)) == 4 and lg[-1] == ("close",)


def test_xr_leg_reports_a_failure_on_every_rank_without_hanging():
    for kw in ({"fail_join": 0}, {"fail_create": 1}, {"fail_rtr": 0}, {"fail_rtr": 2}):
        out, logs = _run(2, leg="xr", **kw)
        for o in out:
            assert "error" in o and "value" not in o
