#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof4mh; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for p in 8 64; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p$p" -- python3 "$ROOT/tools/multiblock_hess_probe.py" 100 20 $p > "$OUT/p$p.log" 2>&1
for f in $(find "$OUT/p$p" -name "*kernel_stats.csv"); do cp "$f" "$OUT/p${p}_kernel_stats.csv"; done
echo "== p=$p"; head -9 "$OUT/p${p}_kernel_stats.csv" | cut -c1-150
done
