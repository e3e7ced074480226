#!/bin/bash
# PMC of the stand-alone S*U kernel at n = 10^6 for one traversal mode: bash tools/gpu_round4_k.sh <sweep>
ROOT=$(pwd); SW=${1:-3}
OUT=$ROOT/gpurun_out/prof4k; rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/h_$c" -- python3 "$ROOT/tools/hess_large_probe.py" 1000 32 --sweep=$SW > "$OUT/h_$c.log" 2>&1
  echo "pmc $c rc=$?"
done
cd "$ROOT"
python3 tools/pmc_to_json.py k_hess_ "$OUT/pmc_hess_n1e6_p32_sweep$SW.json" "$OUT/h_FETCH_SIZE" "$OUT/h_WRITE_SIZE"
cat "$OUT/pmc_hess_n1e6_p32_sweep$SW.json"
rm -rf "$OUT/h_FETCH_SIZE" "$OUT/h_WRITE_SIZE"
