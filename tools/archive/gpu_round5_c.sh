#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/early_probe.py 32,16 > gpurun_out/c_early_probe.log 2>&1
timeout 300 python tools/persist_timeline.py 32 33 > gpurun_out/c_timeline_early33.log 2>&1
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py -x -q 2>&1 | tail -5 > gpurun_out/c_tests_persist.log
for m in 0 4 5 6 7; do MSDP_UC_POOL=$m timeout 600 python tools/uc_pool_stress.py 300 2>&1 | tail -4; done > gpurun_out/c_uc_stress.log 2>&1
cat gpurun_out/c_early_probe.log gpurun_out/c_tests_persist.log gpurun_out/c_uc_stress.log
