bash tools/profile_round4.sh bench 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r4_bench_line.json 2> gpurun_out/r4_bench.err; tail -c 300 gpurun_out/r4_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_line.json"))
print("value", d["value"], "trip", d["tcg_trip_us"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"], "kkt", d.get("g81_kkt",{}).get("seconds_to_dinf_1e-8"))
for e in d.get("dense_mfma",[]): print("dense", e.get("n"), e.get("p"), round(e.get("hessvec_us",0),1), round(e.get("frac_hbm_peak",0),3), round(e.get("frac_mfma_f64_peak",0),3), e.get("roofline",{}).get("traffic"))
for e in d.get("affine_hessvec",[]): print("affine", e.get("workload"), e.get("hessvec_us"), e.get("roofline",{}).get("frac"), e.get("roofline",{}).get("traffic"))
print("large", d.get("large_sparse_trip",{}).get("trip_us"), d.get("large_sparse_trip",{}).get("roofline",{}).get("frac"))
print("xrank", d.get("cross_rank_trip",{}).get("trip_us_cross_rank_persistent"), d.get("cross_rank_trip",{}).get("trip_us_lockstep_chunks"))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
