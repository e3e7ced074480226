"""Where the time of a persistent tCG trip goes: phase stamps (s_memtime, thread 0 of every workgroup) of the traced instance of
k_tcg_persist_obl on G81 -> a table per phase.  Round 5: both trip forms -- persist_early = 0 (gather at the top of the trip,
seven phases) and the EARLY trip (gather behind row flags during reduction 2, eight phases).
and the one-reduction trip (persist_pipe, msdp_pipe.h: five phases).
Writes gpurun_out/r5_persist_timeline_p<p>[_early<k>|_pipe].md.   argv: [p = 32] [persist_early = 1] [persist_pipe = 0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from manisdp_matlab_amd import _lib, problems
p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
early = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
C = problems.maxcut_cost_matrix(os.path.join(ROOT, "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
h = _lib.Handle.onlyunitdiag(C, pcap=p)
h.set_option("persist_early", early)
h.set_option("persist_pipe", pipe)
h.set_point(Y)
plain = min(h.bench_tcg_trip(512) for _ in range(3)) * 1e3
a, j0, ms = h.persist_trace(256)
h.close()
G, nj, _ = a.shape
if pipe:
    # stamps 0, 1, 4, 2, 3, 7, 5, 6 of the traced instance (2, 3, 7 are taken inside the reduction by thread 0; 7 carries wave 0's failed polls in its low 4 bits)
    polls = (a[:, :, 7] & 15).astype(np.float64)
    a = a.copy(); a[:, :, 7] >>= 4
    a = a[:, :, [0, 1, 4, 2, 3, 7, 5, 6]]
    names = ["gather of the neighbours' rows of Hmd, C*tangent(r') and C*mdelta' by linearity, Hmd, its rows stored, eight partial sums (tCG.m:163)",
             "reduction: the eight partial sums over the wave (joint butterfly), to LDS",
             "reduction: wait for the row stores (s_waitcnt vmcnt(0)) + workgroup barrier",
             "reduction: sum over the waves, post, back-off sleep",
             "reduction: wave 0 polls the lines of its 32 workgroups until they have posted (failed polls of wave 0 per trip: %.2f)" % polls.mean(),
             "reduction: sum over the lanes of a value pair, workgroup barrier (= the slowest polling wave), next gather requested, sum over the waves",
             "alpha, tests, commit, beta, new direction (:170-287)",
             "loop back (next trip's set-up)"]
    NP = 8
elif early:
    names = ["top of the trip: C*mdelta' = C*tangent(r') + beta*C*mdelta, projection, <d,Hd> partials (tCG.m:163)",
             "grid reduction 1: <d,Hd> (:166)",
             "trial step, projected residual rows stored, reduction 2 POSTED (:215-241)",
             "last trip's half back to the sentinel, back-off before the first gather",
             "gather of the neighbours' rows until none holds the sentinel, C*tangent(r') formed (reduction 2 polled under the same waits)",
             "rest of grid reduction 2: model value, <r,r> (:227-241)",
             "commit, beta, new direction (:233-287)",
             "loop back (stop tests, next trip's set-up)"]
    NP = 8
else:
    names = ["gathers of C*x + row arithmetic (tCG.m:163)", "grid reduction 1: <d,Hd> (:166)", "trial step, projected residual rows stored (:215-241)",
             "(the wait for those stores sits inside reduction 2 since round 5, behind its wave sums)", "grid reduction 2: model value, <r,r> (:227-241), incl. the drain of the row stores", "commit, beta, new direction (:233-287)",
             "loop back (stop tests, next trip's set-up)"]
    NP = 7
st = a[:, :, :NP].astype(np.float64)
dur = np.empty((G, nj - 1, NP))
dur[:, :, :NP - 1] = st[:, :-1, 1:NP] - st[:, :-1, 0:NP - 1]
dur[:, :, NP - 1] = st[:, 1:, 0] - st[:, :-1, NP - 1]
# a refresh trip (every 32nd) has another shape: leave out the trips whose stamps are not monotone
okt = np.all(dur > 0, axis=(0, 2))
dur = dur[:, okt, :]
trip_ticks = (st[:, 1:, 0] - st[:, :-1, 0])[:, okt].mean()
ns = ms * 1e6 / trip_ticks                       # ns per tick, calibrated on the trip time of the same launch (HIP events)
per_wg = dur.mean(axis=1) * ns                   # [G, NP] ns
lines = []
lines.append("# Persistent tCG trip, phase by phase (G81, n = %d, p = %d, %d workgroups, persist_early = %d)\n" % (n, p, G, early) + (" -- ONE-REDUCTION TRIP (persist_pipe)" if pipe else ""))
lines.append("Trip time of the traced launch: %.3f us (HIP events over 256 trips); the production instance in the same process: %.3f us." % (ms * 1e3, plain))
lines.append("s_memtime tick = %.3f ns (calibrated: %.1f ticks per trip).  Stamps by thread 0 of every workgroup, %d of the trips %d..%d, averaged.\n" % (ns, trip_ticks, int(okt.sum()), j0, j0 + nj - 2))
lines.append("| phase | workgroup 0 | median workgroup | slowest workgroup of the phase | min over workgroups | share of the trip (mean) |")
lines.append("|---|---|---|---|---|---|")
tot = per_wg.sum(axis=1).mean()
for k in range(NP):
    col = per_wg[:, k]
    lines.append("| %d %s | %.0f ns | %.0f ns | %.0f ns (wg %d) | %.0f ns | %.1f %% |" % (k, names[k], col[0], np.median(col), col.max(), int(col.argmax()), col.min(), 100 * col.mean() / tot))
lines.append("| sum | %.0f ns | | | | |" % per_wg[0].sum())
xcd = np.arange(G) % 8
lines.append("\n(s_memtime is not one clock for the chip: stamps of different workgroups are not compared, only differences inside a workgroup.)")
for k in range(NP):
    lines.append("Per-XCD mean of phase %d (workgroup b runs on XCD b mod 8): " % k + ", ".join("%.0f" % per_wg[xcd == x, k].mean() for x in range(8)) + " ns")
# the reduction itself = the wait of the LAST workgroup to arrive (it finds all other slots filled)
r1 = dur[:, :, 1].min(axis=0).mean() * ns
lines.append("Cost of reduction 1 proper = the SHORTEST wait among the workgroups of a trip (the one that arrives last finds the other slots filled): %.0f ns" % r1)
out = "\n".join(lines) + "\n"
print(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", ("r5_persist_timeline_p%d_pipe.md" % p) if pipe else "r5_persist_timeline_p%d_early%d.md" % (p, early)), "w").write(out)
