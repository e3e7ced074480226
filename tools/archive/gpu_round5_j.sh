#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/p_sweep_probe.py 2,4,6,8,10,16 > gpurun_out/j_psweep.log 2>&1; cat gpurun_out/j_psweep.log
timeout 1500 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_edge_cases.py tests/test_gpu_certificates.py tests/test_gpu_known_answers.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dense --no-affine --no-large-sparse --no-xrank 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['hessvec_by_p'], d.get('g81_kkt'))"
