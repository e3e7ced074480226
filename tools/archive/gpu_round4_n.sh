#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_local_ranks.py tests/test_gpu_baseline_sizes.py tests/test_gpu_edge_cases.py -q -x 2>&1 | tail -4 > gpurun_out/n_tests.log
cat gpurun_out/n_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-dense --no-large-sparse --no-affine 2>gpurun_out/n_bench.err > gpurun_out/n_bench.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/n_bench.json').read().strip().splitlines()[-1])
print(d["value"], d["tcg_trip_us"], d.get("g81_kkt",{}).get("seconds_to_dinf_1e-8"))
c=d.get("cross_rank_trip") or {}; print(c.get("trip_us_cross_rank_persistent"), c.get("trip_us_one_unsharded_handle"))
PY
