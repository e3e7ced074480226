"""Average the PMC values of tools/dense_pmc.sh per kernel."""
import csv, glob, collections
for tag in "ab":
    tr = {}
    for f in glob.glob("gpurun_out/dpmc/%s/**/*kernel_trace.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            tr[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/dpmc/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_dense_partial" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] in tr:
                dur[k].append(tr[r["Dispatch_Id"]])
    for k in acc:
        print(tag, k[:40], "avg_ns=%.0f" % (sum(dur[k]) / max(1, len(dur[k]))), {c: sum(v) / len(v) for c, v in acc[k].items()})
