#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ipc_ranks.py -x -q -s 2>&1 | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kkt --no-dense --no-affine --no-large-sparse 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('process_rank_trip'))"
