#!/bin/bash
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x --timeout=1200 2>&1 | tail -15 > gpurun_out/l_fullsuite.log
cat gpurun_out/l_fullsuite.log
