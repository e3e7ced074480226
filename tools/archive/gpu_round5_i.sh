#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/p_sweep_probe.py > gpurun_out/i_psweep.log 2>&1; cat gpurun_out/i_psweep.log
timeout 900 python -m pytest tests/test_gpu_generic.py -x -q -k "sensor" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py -x -q -k "lds_staged" 2>&1 | tail -5
cd examples && timeout 300 python example_snl.py 10 2>&1 | tail -3
