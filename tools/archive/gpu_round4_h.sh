timeout 600 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_edge_cases.py tests/test_gpu_local_ranks.py tests/test_gpu_certificates.py tests/test_gpu_known_answers.py -q 2>&1 | grep -E "passed|failed|FAILED"
timeout 100 python tools/persist_timeline.py 32 2>&1 | tail -22
timeout 100 python tools/xr_trip_probe.py 2 32 2>&1 | tail -3
timeout 600 python bench.py --no-dense --no-affine --no-large-sparse > gpurun_out/r4_bench_short.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_short.json"))
print("value", d["value"], "trip", d["tcg_trip_us"], "frac", d["roofline"]["frac"], "kkt", d.get("g81_kkt",{}).get("seconds_to_dinf_1e-8"), d.get("hessvec_by_p"))
print(d.get("cross_rank_trip"))
PY
