#!/bin/bash
# usage: tools/prof_stats.sh <tag> python3 script args...   -> gpurun_out/prof_<tag>/kernel_stats.csv (plain launches, no graphs)
tag=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
args=()
for a in "$@"; do case "$a" in tools/*|bench.py) args+=("$ROOT/$a");; *) args+=("$a");; esac; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- "${args[@]}" > "$OUT/log.txt" 2>&1
echo "prof $tag rc=$?"
cd "$ROOT"
for f in $(find "$OUT/raw" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; done
head -12 "$OUT/kernel_stats.csv" | cut -c1-200
