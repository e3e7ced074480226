#!/bin/bash
# dense Hess-vec timings for the shapes quoted in DESIGN.md (n=5000 p=32/64, n=20000 p=32, K5 shard)
cd "$(dirname "$0")/.."
timeout 120 python tools/dense_times.py 5000 32 64 20 48 128
timeout 120 python tools/dense_times.py 20000 32
timeout 200 python tools/k5_shard_times.py
