#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py -x -q -k "lds_staged" 2>&1 | tail -8 > gpurun_out/g_tests_window.log
cat gpurun_out/g_tests_window.log
timeout 900 python tools/hess_large_probe.py 1000 32 16 --sweep=3 --window=0,2 --winlds=144,96 > gpurun_out/g_window_probe.log 2>&1
cat gpurun_out/g_window_probe.log
timeout 600 python -m pytest tests/test_gpu_ipc_ranks.py tests/test_gpu_multiblock.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kkt --no-dense --no-affine --no-large-sparse > gpurun_out/g_bench_small.json 2> gpurun_out/g_bench_small.err; tail -c 3000 gpurun_out/g_bench_small.json
