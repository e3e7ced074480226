#!/bin/bash
# Shader clock and power while a dense fp64 contraction runs: rocm-smi sampled beside tools/dense_probe-like load.
mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_samples.log &
SPID=$!
timeout 120 python tools/densesym_probe.py 20000 32 64 > gpurun_out/clock_load.log 2>&1
wait $SPID
sort gpurun_out/clock_samples.log | uniq -c | sort -rn | head -12
tail -4 gpurun_out/clock_load.log
