#!/bin/bash
# average duration of the TR-iteration tail kernel in the bench workload (rocprofv3 kernel stats)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tail; rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kkt > /dev/null 2>&1
grep "k_tr_tail\|k_tcg_persist" "$OUT"/*/*kernel_stats.csv | cut -c1-200
