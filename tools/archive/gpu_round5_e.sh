#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ipc_ranks.py -x -q -s 2>&1 | tail -30 > gpurun_out/e_tests_ipc.log
cat gpurun_out/e_tests_ipc.log
timeout 300 python tools/psync_backoff_probe.py 32 > gpurun_out/e_backoff_p32.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/e_tests_all.log
cat gpurun_out/e_backoff_p32.log gpurun_out/e_tests_all.log
