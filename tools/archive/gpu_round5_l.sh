#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_onlyunitdiag.py -x -q -k "lds_staged" 2>&1 | tail -8
timeout 900 python tools/hess_large_probe.py 1000 32 --sweep=3 --window=0,2,3 --winlds=144
