set -x
bash tools/profile_round4.sh all 2>&1 | tail -40
timeout 600 python bench.py > gpurun_out/r4_bench_line.json 2> gpurun_out/r4_bench.err; tail -c 600 gpurun_out/r4_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_line.json"))
print("value", d["value"], "trip", d["tcg_trip_us"], "frac", d["roofline"]["frac"], "kkt", d.get("g81_kkt",{}).get("seconds_to_dinf_1e-8"))
for e in d.get("dense_mfma",[]): print("dense", e.get("n"), e.get("p"), e.get("hessvec_us"), e.get("frac_hbm_peak"), e.get("frac_mfma_f64_peak"))
for e in d.get("affine_hessvec",[]): print("affine", e.get("workload"), e.get("hessvec_us"), e.get("roofline",{}).get("frac"), e.get("hessvec_us_by_p"))
print("large", {k:v for k,v in d.get("large_sparse_trip",{}).items() if k!="roofline"}, d.get("large_sparse_trip",{}).get("roofline",{}).get("frac"))
print("xrank", d.get("cross_rank_trip"))
print("hess kernel", d["hessvec_kernel"]["kernel_us"], d["hessvec_kernel"]["frac_of_hbm_peak"])
PY
