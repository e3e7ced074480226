"""Reduce rocprofv3 --pmc CSVs (separate FETCH_SIZE / WRITE_SIZE passes) to per-launch HBM bytes of one kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a 16-B-per-lane
coalesced read stream -> doubled; WRITE_SIZE is exact; both are in KiB."""
import csv, glob, json, sys
# usage: pmc_to_json.py <kernel-name-substring> <out.json> [--per N] [--sources a,b,c] <dir>...   (--per: units, e.g. tCG trips, per launch;
# --sources: the kernel's source files, relative to the repository -- their hash goes into the summary so that bench.py can tell a
# summary taken on older kernels from a current one)
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
per = None
sources = []
if "--sources" in argv:
    i = argv.index("--sources"); sources = [x for x in argv[i + 1].split(",") if x]; del argv[i:i + 2]
if "--per" in argv:
    i = argv.index("--per"); per = float(argv[i + 1]); del argv[i:i + 2]
kern, out = argv[0], argv[1]
vals = {}
for d in argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in vals.items()}
res = {"kernel": kern, "counters_avg_per_launch": avg, "launches": {k: len(v) for k, v in vals.items()},
       "fetch_bytes_corrected": 2 * avg.get("FETCH_SIZE", 0) * 1024, "write_bytes": avg.get("WRITE_SIZE", 0) * 1024}
res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
if per:
    res["trips_per_launch"] = per
    res["hbm_bytes_per_trip"] = res["hbm_bytes_per_launch"] / per
if sources:
    hh = hashlib.sha256()
    for f in sources:
        hh.update(open(os.path.join(ROOT, f), "rb").read())
    res["sources"] = sources; res["sources_sha16"] = hh.hexdigest()[:16]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
