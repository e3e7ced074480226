#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run from the repo root):
#   1. --kernel-trace --stats of the bench workload   2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes of a tiny probe
# Every profiled command is the program itself (python3 ...) and is wrapped in `timeout`.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kkt --no-dense > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/pmc_probe.py" 32 > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/pmc_probe.py" 32 > "$OUT/write.log" 2>&1
echo "write rc=$?"
cd "$ROOT"
python3 tools/pmc_to_json.py k_hess_ "$OUT/pmc_hess.json" "$OUT/fetch" "$OUT/write"
python3 tools/pmc_to_json.py k_tcg_persist_obl "$OUT/pmc_persist.json" "$OUT/fetch" "$OUT/write"
for f in $(find "$OUT/stats" -name "*kernel_stats.csv" -o -name "*domain_stats.csv"); do cp "$f" "$OUT/$(basename $f | sed 's/^[0-9]*_//')"; done
ls "$OUT"
