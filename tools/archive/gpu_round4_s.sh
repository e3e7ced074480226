#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/affine_chain_probe.py bqp60 --quick 2>&1 | tee gpurun_out/s_chain.log
timeout 600 python -m pytest tests/test_gpu_affine.py -q -x 2>&1 | tail -3
bash tools/profile_round4.sh affine > gpurun_out/prof4_affine.log 2>&1
head -6 gpurun_out/prof4/bqp60_p32_kernel_stats.csv
python3 - <<'PY'
import json
for f in ("gpurun_out/prof4/pmc_bqp60_p32.json","gpurun_out/prof4/pmc_theta5000_p32.json"):
    d=json.load(open(f))
    print(f)
    for k,v in d["per_kernel_bytes_per_unit"].items():
        print("  %-45s F %.1f MB  W %.1f MB"%(k[:45], v.get("FETCH_SIZE",0)/1e6, v.get("WRITE_SIZE",0)/1e6))
    print("  total", d["hbm_bytes_per_hessvec"]/1e6)
PY
