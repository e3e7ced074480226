#!/bin/bash
# Round-2 rocprofv3 evidence, collected on the GPU box from the repo root (results under gpurun_out/prof2, the
# summaries are then copied into profiles/ by hand).  Every profiled command is `python3 ...` itself under `timeout`;
# counters are collected in their own passes (--kernel-trace + --pmc only).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof2
rm -rf "$OUT"; mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <tag> <timeout> python3 args...
    local tag=$1 to=$2; shift 2
    timeout "$to" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- "$@" > "$OUT/$tag.log" 2>&1
    echo "stats $tag rc=$?"
    for f in $(find "$OUT/$tag" -name "*kernel_stats.csv"); do cp "$f" "$OUT/${tag}_kernel_stats.csv"; done
}
pmc() {     # pmc <tag> <timeout> "<counters>" python3 args...
    local tag=$1 to=$2 ctr=$3; shift 3
    timeout "$to" rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$tag" -- "$@" > "$OUT/$tag.log" 2>&1
    echo "pmc $tag rc=$?"
}
stats bench 300 python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kkt --no-dense --no-affine
pmc fetch 120 FETCH_SIZE python3 "$ROOT/tools/pmc_probe.py" 32
pmc write 120 WRITE_SIZE python3 "$ROOT/tools/pmc_probe.py" 32
stats dense20000 300 python3 "$ROOT/tools/dense_probe.py" 20000 16 32 64
pmc dense20000_fetch 300 FETCH_SIZE python3 "$ROOT/tools/dense_probe.py" 20000 32
pmc dense20000_write 300 WRITE_SIZE python3 "$ROOT/tools/dense_probe.py" 20000 32
pmc dense20000_mfma 300 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" python3 "$ROOT/tools/dense_probe.py" 20000 32 64
stats k5shard 300 python3 "$ROOT/tools/dense_probe.py" 100000 64 --shard 8
stats bqp60_p32 300 python3 "$ROOT/tools/gram_probe.py" 32
stats bqp60_p300 300 python3 "$ROOT/tools/gram_probe.py" 300
stats theta5000 300 python3 "$ROOT/tools/theta_probe.py" 32
stats g81_kkt 300 python3 "$ROOT/tools/g81_escape_profile.py"
stats dual30 300 python3 "$ROOT/tools/dual_probe.py" 30
cd "$ROOT"
python3 tools/pmc_to_json.py k_hess_ "$OUT/pmc_hess_g81_p32.json" "$OUT/fetch" "$OUT/write"
python3 tools/pmc_to_json.py k_tcg_persist_obl "$OUT/pmc_persist_g81_p32.json" --per 64 "$OUT/fetch" "$OUT/write"
python3 tools/pmc_to_json.py k_dense_partial3 "$OUT/pmc_dense20000_p32.json" "$OUT/dense20000_fetch" "$OUT/dense20000_write"
python3 tools/pmc_counters.py "$OUT/dense20000_mfma" k_dense_partial3 > "$OUT/pmc_dense20000_mfma.txt"
ls "$OUT" | grep -v "^[a-z0-9_]*$" 
