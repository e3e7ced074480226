"""Where the time of a persistent tCG trip goes: phase stamps (s_memtime) of every workgroup of the
traced instance of k_tcg_persist_obl on G81, p = 32 -> a table per phase.  Writes gpurun_out/r4_persist_timeline.md.
argv: [p=32] [graph file]"""
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
h.set_point(Y)
plain = min(h.bench_tcg_trip(512) for _ in range(3)) * 1e3
a, j0, ms = h.persist_trace(256)
h.close()
G, nj, _ = a.shape
names = ["gathers of C*x + row arithmetic (tCG.m:163)", "grid reduction 1: <d,Hd> (:166)", "trial step, projected residual rows stored (:215-241)",
         "wait for those stores (s_waitcnt vmcnt(0))", "grid reduction 2: model value, <r,r> (:227-241)", "commit, beta, new direction (:233-287)",
         "loop back (stop tests, next trip's set-up)"]
st = a[:, :, :7].astype(np.float64)
dur = np.empty((G, nj - 1, 7))
dur[:, :, :6] = st[:, :-1, 1:7] - st[:, :-1, 0:6]
dur[:, :, 6] = st[:, 1:, 0] - st[:, :-1, 6]
trip_ticks = (st[:, 1:, 0] - st[:, :-1, 0]).mean()
ns = ms * 1e6 / trip_ticks                       # ns per tick, calibrated on the trip time of the same launch (HIP events)
per_wg = dur.mean(axis=1) * ns                   # [G, 7] ns
lines = []
lines.append("# Persistent tCG trip, phase by phase (G81, n = %d, p = %d, %d workgroups)\n" % (n, p, G))
lines.append("Trip time of the traced launch: %.3f us (HIP events over 256 trips); the production instance in the same process: %.3f us." % (ms * 1e3, plain))
lines.append("s_memtime tick = %.3f ns (calibrated: %.1f ticks per trip).  Stamps by thread 0 of every workgroup, trips %d..%d, averaged.\n" % (ns, trip_ticks, j0, j0 + nj - 2))
lines.append("| phase | workgroup 0 | median workgroup | slowest workgroup of the phase | min over workgroups | share of the trip (mean) |")
lines.append("|---|---|---|---|---|---|")
tot = per_wg.sum(axis=1).mean()
for k in range(7):
    col = per_wg[:, k]
    lines.append("| %d %s | %.0f ns | %.0f ns | %.0f ns (wg %d) | %.0f ns | %.1f %% |" % (k, names[k], col[0], np.median(col), col.max(), int(col.argmax()), col.min(), 100 * col.mean() / tot))
lines.append("| sum | %.0f ns | | | | |" % per_wg[0].sum())
# arrival skew at the two reductions (stamps 1 and 4 are taken right before the post); only meaningful if s_memtime is one clock for the chip
xcd = np.arange(G) % 8
# arrival skew at the two reductions, per XCD (s_memtime is one counter per XCD, not per chip: stamps of different XCDs do not compare)
sk1 = np.mean([(st[xcd == x][:, :, 1].max(axis=0) - st[xcd == x][:, :, 1].min(axis=0)).mean() for x in range(8)]) * ns
sk2 = np.mean([(st[xcd == x][:, :, 4].max(axis=0) - st[xcd == x][:, :, 4].min(axis=0)).mean() for x in range(8)]) * ns
lines.append("\nArrival skew inside an XCD (last minus first of its workgroups to reach the post, mean over trips and XCDs): reduction 1 %.0f ns, reduction 2 %.0f ns" % (sk1, sk2))
lines.append("Per-XCD mean of the gather phase (workgroup b runs on XCD b mod 8): " + ", ".join("%.0f" % per_wg[xcd == x, 0].mean() for x in range(8)) + " ns")
lines.append("Per-XCD mean wait in reduction 1: " + ", ".join("%.0f" % per_wg[xcd == x, 1].mean() for x in range(8)) + " ns")
lines.append("Per-XCD mean wait in reduction 2: " + ", ".join("%.0f" % per_wg[xcd == x, 4].mean() for x in range(8)) + " ns")
# the reduction itself = the wait of the LAST workgroup to arrive (it finds all other slots filled)
r1 = dur[:, :, 1].min(axis=0).mean() * ns; r2 = dur[:, :, 4].min(axis=0).mean() * ns
lines.append("Cost of a reduction proper = the SHORTEST wait among the workgroups of a trip (the one that arrives last finds the other slots filled): reduction 1 %.0f ns, reduction 2 %.0f ns" % (r1, r2))
out = "\n".join(lines) + "\n"
print(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r4_persist_timeline_p%d.md" % p), "w").write(out)
