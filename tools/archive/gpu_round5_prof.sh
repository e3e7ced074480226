#!/bin/bash
# profiles of round 5: kernel stats + PMC passes (tools/profile_round5.sh), the bench line, the timelines of the default trip
mkdir -p gpurun_out
bash tools/profile_round5.sh bench dense sparse > gpurun_out/prof5_log.txt 2>&1
tail -30 gpurun_out/prof5_log.txt
timeout 300 python tools/persist_timeline.py 32 0 > gpurun_out/p_timeline.log 2>&1
timeout 900 python bench.py > gpurun_out/r5_bench_line.json 2> gpurun_out/r5_bench_err.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_line.json"))
print({k: d[k] for k in ("metric", "value", "ms_per_step", "roofline")})
print("g81_kkt", d.get("g81_kkt"))
print("dense", [(e.get("n"), e.get("p"), e.get("hessvec_us"), e.get("roofline", {}).get("bound"), e.get("roofline", {}).get("frac")) for e in d.get("dense_mfma", [])])
print("affine", [(e.get("workload"), e.get("hessvec_us"), e.get("roofline", {}).get("frac")) for e in d.get("affine_hessvec", [])])
print("large", d.get("large_sparse_trip"))
print("cpu", d.get("cpu_baseline"))
PY
