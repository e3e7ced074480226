#!/bin/bash
mkdir -p gpurun_out
timeout 1000 python -m pytest tests/test_gpu_multiblock.py -q -x 2>&1 | tail -4
L=gpurun_out/r_multiblock_device_eig.log; : > $L
cd examples
for args in "20 20" "100 20" "400 10"; do
  timeout 900 python example_bqp_sparse.py $args >> ../$L 2>&1
done
cd ..
timeout 600 python tools/multiblock_breakdown.py 100 20 2>&1 | head -3 >> $L
timeout 600 python tools/multiblock_scale_probe.py >> $L 2>&1
cat $L
