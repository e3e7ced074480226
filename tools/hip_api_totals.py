"""Total host time per HIP API function over the last <ms> milliseconds of a rocprofv3 --hip-trace CSV (the second solve of
tools/kkt_timing_probe.py), longest first, and the calls above 0.3 ms.
    python tools/hip_api_totals.py <dir> <window ms>"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
win = float(sys.argv[2]) * 1e6
for f in glob.glob(os.path.join(d, "**", "*_hip_api_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tend = max(int(r["End_Timestamp"]) for r in rows if "Synchronize" in r["Function"])
    t0 = tend - win
    tot, cnt = defaultdict(int), defaultdict(int)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t0:
            continue
        tot[r["Function"]] += e - s; cnt[r["Function"]] += 1
        if e - s > 300_000 and "Synchronize" not in r["Function"]:
            print("   %.2f ms at %+.2f ms: %s" % ((e - s) / 1e6, (s - t0) / 1e6, r["Function"]))
    for k in sorted(tot, key=lambda k: -tot[k])[:16]:
        print("%-32s %6d calls %9.3f ms" % (k, cnt[k], tot[k] / 1e6))
