#!/bin/bash
# kernel stats of the multiblock solve of 100 cliques (blocks of order 211) with eig(S_i) on the device
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof4mb; rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT/examples:$ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mb" -- python3 "$ROOT/examples/example_bqp_sparse.py" 100 20 > "$OUT/mb.log" 2>&1
for f in $(find "$OUT/mb" -name "*kernel_stats.csv"); do cp "$f" "$OUT/mb_kernel_stats.csv"; done
head -14 "$OUT/mb_kernel_stats.csv" | cut -c1-160
grep ManiSDP "$OUT/mb.log"
