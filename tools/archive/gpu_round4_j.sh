#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multiblock.py -q -x 2>&1 | tail -8 > gpurun_out/j_multiblock.log
cat gpurun_out/j_multiblock.log
timeout 900 python tools/multiblock_scale_probe.py > gpurun_out/j_scale.log 2>&1; cat gpurun_out/j_scale.log
