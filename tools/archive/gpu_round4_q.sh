#!/bin/bash
# kernel stats of one G81 solve to KKT 1e-8 (p0 = 40): where the 0.18 s go
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof4kkt; rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kkt" -- python3 "$ROOT/tools/solve_g81.py" > "$OUT/kkt.log" 2>&1
for f in $(find "$OUT/kkt" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kkt_kernel_stats.csv"; done
head -16 "$OUT/kkt_kernel_stats.csv" | cut -c1-150
tail -3 "$OUT/kkt.log"
