"""Summarise a rocprofv3 kernel trace: per kernel name, runs of consecutive dispatches with their mean duration
(used to see how k_tcg_persist_obl behaves from one RTR call to the next inside a full solve)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "k_tcg_persist"
sel = [(s, e, n) for s, e, n in rows if key in n]
# split into groups separated by gaps > 20 ms
groups, cur = [], []
for s, e, n in sel:
    if cur and s - cur[-1][1] > 20_000_000:
        groups.append(cur); cur = []
    cur.append((s, e, n))
if cur: groups.append(cur)
for g in groups:
    d = [(e - s) / 1e3 for s, e, _ in g]
    span = (g[-1][1] - g[0][0]) / 1e6
    print("%-40s launches %3d  mean %.1f us  max %.1f us  sum %.2f ms  span %.2f ms" % (g[0][2][:40], len(g), sum(d) / len(d), max(d), sum(d) / 1e3, span))
