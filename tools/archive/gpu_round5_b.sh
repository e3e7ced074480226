#!/bin/bash
# round 5, second GPU pass: the sentinel form of the EARLY persistent trip (correctness, timings, timeline) + the IPC probe
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -15 > gpurun_out/b_tests_persist.log
timeout 900 python tools/early_probe.py 32,16,8 > gpurun_out/b_early_probe.log 2>&1
timeout 300 python tools/persist_timeline.py 32 1 > gpurun_out/b_timeline_early1.log 2>&1
timeout 300 python tools/persist_timeline.py 32 0 > gpurun_out/b_timeline_early0.log 2>&1
timeout 400 bash tools/ipc_probe.sh > gpurun_out/b_ipc_probe.log 2>&1
tail -5 gpurun_out/b_tests_persist.log; cat gpurun_out/b_early_probe.log; cat gpurun_out/b_ipc_probe.log
