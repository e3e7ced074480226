#!/bin/bash
# Round-5 rocprofv3 evidence, collected on the GPU box from the repo root (results under gpurun_out/prof5; the summaries are then
# copied into profiles/ as r5_*).  Every profiled command is `python3 ...` itself under `timeout`; counters are collected in their
# own passes (--kernel-trace + --pmc only); graphs are off (rocprofv3 7.2 crashes on graph replay).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof5
mkdir -p "$OUT"
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <tag> <timeout> python3 args...
    local tag=$1 to=$2; shift 2
    timeout "$to" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- "$@" > "$OUT/$tag.log" 2>&1
    echo "stats $tag rc=$?"
    # (child processes of the profiled command write their own files: the parent's is the one with the lowest process id)
    for f in $(find "$OUT/$tag" -name "*kernel_stats.csv" | sort -t/ -k1 -V -r); do cp "$f" "$OUT/${tag}_kernel_stats.csv"; done
}
pmc() {     # pmc <tag> <timeout> "<counters>" python3 args...
    local tag=$1 to=$2 ctr=$3; shift 3
    timeout "$to" rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$tag" -- "$@" > "$OUT/$tag.log" 2>&1
    echo "pmc $tag rc=$?"
}
both() {    # both <tag> <timeout> python3 args...   (FETCH_SIZE and WRITE_SIZE in separate passes)
    local tag=$1 to=$2; shift 2
    pmc ${tag}_fetch $to FETCH_SIZE "$@"
    pmc ${tag}_write $to WRITE_SIZE "$@"
}
WHAT=" ${*:-all} "     # one or more of: bench dense sparse (default: all)
want() { [[ "$WHAT" == *" all "* || "$WHAT" == *" $1 "* ]]; }
if want bench; then
stats bench 400 python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kkt --no-dense --no-affine --no-large-sparse
both persist 120 python3 "$ROOT/tools/pmc_probe.py" 32 0
both pipe 120 python3 "$ROOT/tools/pmc_probe.py" 32 1
fi
if want dense; then
stats dense20000 300 python3 "$ROOT/tools/dense_probe.py" 20000 16 32 64
both dense20000p16 300 python3 "$ROOT/tools/dense_probe.py" 20000 16
both dense20000p32 300 python3 "$ROOT/tools/dense_probe.py" 20000 32
# matrix-pipe utilisation of the symmetric contraction (missing 3): busy cycles of the MFMA pipe against the CU-busy cycles
pmc dense20000_mfma 300 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" python3 "$ROOT/tools/dense_probe.py" 20000 16 32
fi
if want sparse; then
stats hess1e6 300 python3 "$ROOT/tools/hess_large_probe.py" 1000 32 --sweep=3 --window=0,2
both hess1e6win 300 python3 "$ROOT/tools/hess_large_probe.py" 1000 32 --sweep=3 --window=2
fi
cd "$ROOT"
[ -d "$OUT/persist_fetch" ] && python3 tools/pmc_to_json.py k_tcg_persist_obl "$OUT/pmc_persist_g81_p32.json" --per 64 "$OUT/persist_fetch" "$OUT/persist_write"
[ -d "$OUT/pipe_fetch" ] && python3 tools/pmc_to_json.py k_tcg_pipe_obl "$OUT/pmc_pipe_g81_p32.json" --per 64 "$OUT/pipe_fetch" "$OUT/pipe_write"
for p in 16 32; do
  [ -d "$OUT/dense20000p${p}_fetch" ] && python3 tools/pmc_sum.py "$OUT/pmc_dense20000_p$p.json" k_dense_hess_epi hbm_bytes_per_hessvec --only k_dense_sym,k_sym_fold,k_dense_partial3,k_dense_hess_epi "$OUT/dense20000p${p}_fetch" "$OUT/dense20000p${p}_write"
done
[ -d "$OUT/dense20000_mfma" ] && python3 tools/pmc_counters.py "$OUT/dense20000_mfma" k_dense_sym > "$OUT/pmc_dense20000_sym_mfma.txt"
[ -d "$OUT/hess1e6win_fetch" ] && python3 tools/pmc_to_json.py k_hess_win "$OUT/pmc_hess_win_n1e6_p32.json" "$OUT/hess1e6win_fetch" "$OUT/hess1e6win_write"
ls "$OUT" | grep -v "^[a-z0-9_]*$"
