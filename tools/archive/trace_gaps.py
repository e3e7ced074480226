"""List the idle gaps (> threshold ms) of a rocprofv3 kernel trace with the kernels on both sides."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
rows.sort()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
t0 = rows[0][0]
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = (s1 - e0) / 1e6
    if gap > thr:
        print("t=%8.1f ms  gap %7.2f ms  after %-48s (%.0f us)  before %s" % ((e0 - t0) / 1e6, gap, n0, (e0 - s0) / 1e3, n1))
