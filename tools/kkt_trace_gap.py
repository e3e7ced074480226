"""Reads a rocprofv3 --kernel-trace --memory-copy-trace CSV pair of tools/kkt_timing_probe.py and prints what the device did
in the 40 ms before the first kernel of the LAST solve's first trustregions() (the wait msdp_set_point reports).
    python tools/kkt_trace_gap.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
t0 = ev[0][0]
# gaps > 3 ms in the device timeline
last_end = ev[0][1]
for i, (s, e, nme) in enumerate(ev):
    if s - last_end > 3_000_000:
        print("gap %.2f ms before event %d at %.2f ms: %s" % ((s - last_end) / 1e6, i, (s - t0) / 1e6, nme))
        for s2, e2, n2 in ev[max(0, i - 4):i + 6]:
            print("     %10.3f ms  %9.3f ms  %s" % ((s2 - t0) / 1e6, (e2 - s2) / 1e6, n2))
    last_end = max(last_end, e)
# long events
for s, e, nme in ev:
    if e - s > 2_000_000:
        print("long %.2f ms at %.2f ms: %s" % ((e - s) / 1e6, (s - t0) / 1e6, nme))
