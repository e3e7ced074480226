"""Average every PMC counter of one rocprofv3 --pmc pass per kernel (name substring argv[2]), with the kernel durations."""
import csv, glob, collections, sys
d, kern = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dd = collections.defaultdict(list)
seen = set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kern not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dd[k].append(dur[r["Dispatch_Id"]])
for k in acc:
    print(k[:60], "launches", len(dd[k]), "avg_ns %.0f" % (sum(dd[k]) / max(1, len(dd[k]))))
    for c, v in sorted(acc[k].items()):
        print("   %-36s %.6g" % (c, sum(v) / len(v)))
