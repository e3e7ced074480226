#!/bin/bash
# round 5, batch n: full GPU suite with the one-reduction trip as default, default bench line, profiles of the pipe kernel
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/n_tests.log
timeout 900 python bench.py > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
tail -c 600 gpurun_out/n_bench.err
bash tools/profile_round5.sh bench > gpurun_out/n_prof.log 2>&1
timeout 300 python tools/persist_timeline.py 32 0 1 > gpurun_out/n_timeline.log 2>&1
cat gpurun_out/n_tests.log
python - <<'PY'
import json
l = [x for x in open("gpurun_out/n_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
print(d["value"], d["tcg_trip_us"], d["roofline"]["frac"], d["roofline"].get("two_reduction_trip_us"), d["roofline"]["kernel"])
for k in ("kkt_solve", "g81_kkt", "kkt"):
    if k in d: print(k, d[k])
print({k: v for k, v in d.items() if "kkt" in k.lower()})
PY
