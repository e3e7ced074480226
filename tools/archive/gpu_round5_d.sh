#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/early_probe.py 32,16 > gpurun_out/d_early_probe.log 2>&1
timeout 300 python tools/persist_timeline.py 32 9 > gpurun_out/d_timeline_early9.log 2>&1
timeout 300 python tools/persist_timeline.py 32 17 > gpurun_out/d_timeline_early17.log 2>&1
MSDP_UC_POOL=7 timeout 600 python tools/early_probe.py 32 > gpurun_out/d_early_probe_finegrained.log 2>&1
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py -x -q 2>&1 | tail -5 > gpurun_out/d_tests_persist.log
cat gpurun_out/d_early_probe.log gpurun_out/d_tests_persist.log; echo FINEGRAINED; cat gpurun_out/d_early_probe_finegrained.log
