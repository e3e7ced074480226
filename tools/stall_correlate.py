"""For every HIP API call longer than <ms> in a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace CSV set: the device-side
events (kernels, copies) that started or ended inside the call -- was the device busy, late, or idle while the host waited?
    python tools/stall_correlate.py <dir> [ms] [name filter]"""
import csv, glob, os, sys
d = sys.argv[1]
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
flt = sys.argv[3] if len(sys.argv) > 3 else "Synchronize"
def load(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
api = load("*_hip_api_trace.csv")
dev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]) for r in load("*_kernel_trace.csv")]
dev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", "")) for r in load("*_memory_copy_trace.csv")]
dev.sort()
api.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(api[0]["Start_Timestamp"])
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < thr or flt not in r["Function"]:
        continue
    inside = [x for x in dev if x[1] >= s and x[0] <= e]
    busy = sum(min(x[1], e) - max(x[0], s) for x in inside)
    print("%s %.2f ms at %.2f ms: %d device events overlap, device busy %.2f ms of it" % (r["Function"], (e - s) / 1e6, (s - t0) / 1e6, len(inside), busy / 1e6))
    for x in inside[:3] + (inside[-3:] if len(inside) > 3 else []):
        print("      %+9.3f ms (from call start) for %.3f ms  %s" % ((x[0] - s) / 1e6, (x[1] - x[0]) / 1e6, x[2]))
