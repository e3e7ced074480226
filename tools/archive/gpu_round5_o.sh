#!/bin/bash
# round 5, batch o: kernel stats of whole trustregions() calls at p = 40 (per-iteration launches: tCG kernel + TR tail kernel)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof5o; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p40" -- python3 "$ROOT/tools/pipe_p40_probe.py" 40 > "$OUT/p40.log" 2>&1
for f in $(find "$OUT/p40" -name "*kernel_stats.csv"); do head -8 "$f" | cut -c1-170; done
