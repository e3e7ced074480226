#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/goff_probe.py 32,16,8 > gpurun_out/h_goff.log 2>&1; cat gpurun_out/h_goff.log
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_ipc_ranks.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kkt --no-dense --no-affine --no-large-sparse 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('process_rank_trip'))"
