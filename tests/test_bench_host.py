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
        self.log.append(("rtr",))
        return _FakeStats()

    def close(self):
        self.log.append(("close",))


def _run(N, fail_create=None, fail_join=None):
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
                    return _FakeHandle(logs[r])

        def join(h):
            if r == fail_join:
                raise RuntimeError("communicator set-up failed")

        def allmax(x):
            box[r] = x
            bar.wait()
            m = max(box)
            bar.wait()
            return m

        out[r] = bench.halo_leg(Lib, join, bar.wait, allmax, N, r, None, np.zeros((4, 2)), 2, None, 3, 1)

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
