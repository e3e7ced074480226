"""Round 6: where a TR iteration of the FUSED launch spends its time outside the tCG trips (msdp_debug_persist_trace with reps = 0;
FSTAMP in msdp_pipe.h).  G81, seed-0 start point, maxiter = 40, maxinner = 100 -- bench.py's step.
usage: python tools/fused_timeline.py [p] [out.md]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
h.set_point(Y)
st = h.rtr(opts)                       # (sets the handle's last options; warms everything up)
trip_ms = None
h.set_point(Y)
trip_ms = h.bench_tcg_trip(512)
# the tick of the stamps (the shader clock): calibrated on the per-iteration instance's traced launch, whose trip time HIP events give
h.set_option("fused_rtr", 0)
h.set_point(Y)
a2, _, ms2 = h.persist_trace(256)
d0 = np.diff(a2[:, :, 0], axis=1)
tick_us = ms2 * 1e3 / float(np.median(d0))
h.set_option("fused_rtr", 1)
h.set_point(Y)
h.rtr(opts)
h.set_point(Y)
(a, trip_a), j0, call_ms = h.persist_trace(0)
G, nj, _ = a.shape
mask = (1 << 56) - 1
trips = (a[:, :, 2] >> 56) & 0xff
t = a & mask
its = [k for k in range(nj) if (t[:, k, 7] > 0).all()]
# tick: calibrated on the whole call -- first stamp 0 to last stamp 7 of workgroup 0 against the call's kernel time is not available;
# use the trip time of the non-fused instance on the tCG phases instead: ticks per trip from (stamp 2 - stamp 1) / (trips - 1)
tick_samples = []
for k in its:
    tr = int(np.median(trips[:, k]))
    if tr >= 8:
        tick_samples.append(np.median(t[:, k, 2] - t[:, k, 1]) / (tr - 1))
names = ["iteration start -> first trip's products formed (tCG.m:102-163: reset + first gather)", "... -> tCG ended (the remaining trips)",
         "retraction, proposal rows stored and performed (trustregions.m:540)", "barrier", "gather of the proposal rows, cost / gradient rows (:544)",
         "the iteration's reduction", "decision, point committed (:548-729)"]
lines = []
lines.append("# TR iteration of the fused launch outside its trips (G81, p = %d, %d workgroups; %d iterations stamped)\n" % (p, G, len(its)))
lines.append("call: %.3f ms for %d Hess-vecs in %d iterations = %.2f us per Hess-vec; trip of the per-iteration instance in this run: %.2f us\n"
             % (call_ms, st.hessvecs, st.iters, call_ms * 1e3 / st.hessvecs, trip_ms * 1e3))
ticks_per_trip = float(np.median(tick_samples)) if tick_samples else float("nan")
lines.append("median ticks per fused trip (stamps 1 -> 2 over trips - 1): %.0f\n" % ticks_per_trip)
# absolute scale: the whole call in ticks (workgroup 0: first stamp 0 -> last stamp 7) against call_ms would include host time; instead
# report ticks and microseconds under the assumption of a 100 MHz s_memtime clock, and the ratio to a fused trip
ph = np.zeros((len(its), 7))
for i, k in enumerate(its):
    for q in range(7):
        ph[i, q] = np.median(t[:, k, q + 1] - t[:, k, q])
tot = t[0, its[-1], 7] - t[0, its[0], 0]
hv = int(sum(int(np.median(trips[:, k])) for k in its))
lines.append("stamped span: %d ticks for %d Hess-vecs\n" % (tot, hv))
# ticks -> us: the stamped span of workgroup 0 covers its[0]..its[-1]; the call's device time is ~ call_ms minus host overhead, so calibrate
# on sum over iterations = span and span_us = (fraction of Hess-vecs stamped) * kernel time is circular; use the constant-frequency
# counter instead: s_memtime runs at 100 MHz on gfx950
us = lambda x: x * tick_us
lines.append("| phase | median over iterations (ticks) | us | in fused trips |")
lines.append("|---|---|---|---|")
for q in range(7):
    if q == 1:
        continue
    m = float(np.median(ph[:, q]))
    lines.append("| %s | %.0f | %.2f | %.2f |" % (names[q], m, us(m), m / ticks_per_trip))
outside = ph[:, [0, 2, 3, 4, 5, 6]].sum(axis=1)
lines.append("| **sum outside the trips 2..j** | %.0f | %.2f | %.2f |" % (np.median(outside), us(np.median(outside)), np.median(outside) / ticks_per_trip))
lines.append("\nfused trip: %.2f us; per-iteration table (ticks): iteration, trips, phases 0..6" % us(ticks_per_trip))
for i, k in enumerate(its[:12]):
    lines.append("  %2d %3d  " % (k, int(np.median(trips[:, k]))) + " ".join("%6.0f" % x for x in ph[i]))
# the trips of ONE TR iteration (the phases of tools/persist_timeline.py: 0 top, 1 products + partial sums formed, 4 wave butterfly done,
# 2 stores performed + workgroup barrier, 3 posted + slept, 7 wave 0's poll returned, 5 reduction returned, 6 new direction formed), and
# the same table for the per-iteration instance's traced launch of this run
def trip_table(tt, title):
    valid = [jj for jj in range(tt.shape[1] - 1) if (tt[:, jj, 0] > 0).all() and (tt[:, jj + 1, 0] > 0).all() and (tt[:, jj, 6] > 0).all()]
    if not valid:
        return
    lines.append("\n%s (%d trips stamped), median over workgroups and trips, us:" % (title, len(valid)))
    seq = [("top -> products, Hmd rows stored, partial sums", 0, 1), ("wave butterfly", 1, 4), ("store drain + workgroup barrier", 4, 2),
           ("cross-wave sum, post, back-off", 2, 3), ("poll (wave 0)", 3, 7), ("lane sums, barrier, results (+ next gather issued)", 7, 5),
           ("alpha .. new direction", 5, 6)]
    tot = 0.0
    for nm, p0, p1 in seq:
        v1 = tt[:, valid, p1] >> 4 if p1 == 7 else tt[:, valid, p1]
        v0 = tt[:, valid, p0] >> 4 if p0 == 7 else tt[:, valid, p0]
        m = float(np.median(v1 - v0)); tot += m
        lines.append("  %-55s %6.2f" % (nm, us(m)))
    nxt = [jj + 1 for jj in valid]
    loop = float(np.median(tt[:, nxt, 0] - tt[:, valid, 6]))
    lines.append("  %-55s %6.2f" % ("loop back", us(loop)))
    lines.append("  %-55s %6.2f   (trip to trip, stamp 0: %.2f)" % ("sum", us(tot + loop), us(float(np.median(tt[:, nxt, 0] - tt[:, valid, 0])))))
trip_table(trip_a, "trips of TR iteration 7 of the fused launch")
trip_table(a2, "trips of the per-iteration instance (bench mode, 256 trips in one launch)")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
h.close()
