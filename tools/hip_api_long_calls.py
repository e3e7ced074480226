"""Lists the HIP API calls of a rocprofv3 --hip-trace CSV that took longer than a threshold, with the calls just before each one.
    python tools/hip_api_long_calls.py <dir> [ms]"""
import csv, glob, os, sys
d = sys.argv[1]
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
for f in glob.glob(os.path.join(d, "**", "*_hip_api_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for i, r in enumerate(rows):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur > thr:
            print("%.2f ms at %.2f ms: %s (thread %s)" % (dur / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"], r.get("Thread_Id", "")))
            for q in rows[max(0, i - 6):i]:
                print("      before: %10.3f ms %8.3f ms %s" % ((int(q["Start_Timestamp"]) - t0) / 1e6, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e6, q["Function"]))
