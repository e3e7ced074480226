#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ipc_ranks.py -x -q -s 2>&1 | tail -30 > gpurun_out/f_tests_ipc.log
cat gpurun_out/f_tests_ipc.log
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py -x -q -k "lds_staged" 2>&1 | tail -15 > gpurun_out/f_tests_window.log
cat gpurun_out/f_tests_window.log
timeout 900 python tools/hess_large_probe.py 1000 32 16 --sweep=3 --window=0,2 --winlds=144,96,72,48 > gpurun_out/f_window_probe.log 2>&1
cat gpurun_out/f_window_probe.log
timeout 600 python -m pytest tests/test_gpu_local_ranks.py -x -q -k "cross_rank" 2>&1 | tail -5
