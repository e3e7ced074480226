"""HBM bytes per work unit of a CHAIN of kernels (an affine Hess-vec is five launches) from rocprofv3 --pmc CSVs
(separate FETCH_SIZE / WRITE_SIZE passes; gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports half of the bytes of a
16-B-per-lane coalesced read stream -> doubled; WRITE_SIZE is exact; both in KiB).
usage: pmc_sum.py <out.json> <units: launches of the kernel whose name contains this string> <field> [--only a,b,c] <dir> [<dir> ...]
The dispatches of the kernels whose names contain one of the --only substrings (default: every dispatch of the run) are summed
and divided by the number of dispatches of the unit kernel; the per-kernel split is kept in the record.  The x2 of FETCH_SIZE
is the guide's correction for 16-B-per-lane streams; for the 8-byte gathers of the affine operators it is an upper bound."""
import csv, glob, json, sys, collections
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sources = []
if "--sources" in sys.argv:            # the kernels' source files (relative to the repository): their hash goes into the summary
    i = sys.argv.index("--sources"); sources = [x for x in sys.argv[i + 1].split(",") if x]; del sys.argv[i:i + 2]
out, unit_kernel, field = sys.argv[1], sys.argv[2], sys.argv[3]
rest = sys.argv[4:]
only = None
if rest and rest[0] == "--only":
    only = rest[1].split(","); rest = rest[2:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in rest:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:80]
            if only is not None and not any(o in k for o in only):
                continue
            tot[r["Counter_Name"]][k] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]][k] += 1
units = {c: sum(n for k, n in cnt[c].items() if unit_kernel in k) for c in cnt}
rec = {"unit_kernel": unit_kernel, "units": units, "kernels_included": only, "per_kernel_bytes_per_unit": {}}
total = 0.0
for c, scale in (("FETCH_SIZE", 2 * 1024.0), ("WRITE_SIZE", 1024.0)):
    u = max(1, units.get(c, 0))
    for k, v in tot[c].items():
        b = v * scale / u
        rec["per_kernel_bytes_per_unit"].setdefault(k, {})[c] = b
        total += b
rec[field] = total
if sources:
    hh = hashlib.sha256()
    for f in sources:
        hh.update(open(os.path.join(ROOT, f), "rb").read())
    rec["sources"] = sources; rec["sources_sha16"] = hh.hexdigest()[:16]
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("unit_kernel", "units", field)}))
