#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof4q; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MSDP_NO_GRAPH=1
export PYTHONPATH=$ROOT/examples:$ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/q" -- python3 "$ROOT/examples/example_rotationsearch.py" 200 > "$OUT/q.log" 2>&1
for f in $(find "$OUT/q" -name "*kernel_stats.csv"); do cp "$f" "$OUT/q_kernel_stats.csv"; done
head -12 "$OUT/q_kernel_stats.csv" | cut -c1-170
grep -i "ManiSDP\|QUASAR\|rank" "$OUT/q.log"
