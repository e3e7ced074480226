#!/bin/bash
# Round-4 rocprofv3 evidence, collected on the GPU box from the repo root (results under gpurun_out/prof4; the summaries are then
# copied into profiles/ as r4_*).  Every profiled command is `python3 ...` itself under `timeout`; counters are collected in their
# own passes (--kernel-trace + --pmc only); graphs are off (rocprofv3 7.2 crashes on graph replay).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof4
mkdir -p "$OUT"
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
both() {    # both <tag> <timeout> python3 args...   (FETCH_SIZE and WRITE_SIZE in separate passes)
    local tag=$1 to=$2; shift 2
    pmc ${tag}_fetch $to FETCH_SIZE "$@"
    pmc ${tag}_write $to WRITE_SIZE "$@"
}
WHAT=" ${*:-all} "     # one or more of: bench dense affine sparse (default: all)
want() { [[ "$WHAT" == *" all "* || "$WHAT" == *" $1 "* ]]; }
if want bench; then
stats bench 400 python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kkt --no-dense --no-affine --no-large-sparse
both persist 120 python3 "$ROOT/tools/pmc_probe.py" 32
fi
if want dense; then
stats dense20000 300 python3 "$ROOT/tools/dense_probe.py" 20000 16 32 64
both dense20000p16 300 python3 "$ROOT/tools/dense_probe.py" 20000 16
both dense20000p32 300 python3 "$ROOT/tools/dense_probe.py" 20000 32
fi
if want affine; then
stats bqp60_p32 300 python3 "$ROOT/tools/gram_probe.py" 32
both bqp60 300 python3 "$ROOT/tools/gram_probe.py" 32
stats theta5000 300 python3 "$ROOT/tools/theta_probe.py" 32
both theta5000 300 python3 "$ROOT/tools/theta_probe.py" 32
fi
if want sparse; then
stats hess1e6 300 python3 "$ROOT/tools/hess_large_probe.py" 1000 32 --sweep=1
both hess1e6 300 python3 "$ROOT/tools/hess_large_probe.py" 1000 32 --sweep=1
stats linear1e6 300 python3 "$ROOT/tools/pmc_probe_large.py" 1000 1000 32
fi
cd "$ROOT"
[ -d "$OUT/persist_fetch" ] && python3 tools/pmc_to_json.py k_tcg_persist_obl "$OUT/pmc_persist_g81_p32.json" --per 64 "$OUT/persist_fetch" "$OUT/persist_write"
[ -d "$OUT/persist_fetch" ] && python3 tools/pmc_to_json.py k_hess_ "$OUT/pmc_hess_g81_p32.json" "$OUT/persist_fetch" "$OUT/persist_write"
for p in 16 32; do
  [ -d "$OUT/dense20000p${p}_fetch" ] && python3 tools/pmc_sum.py "$OUT/pmc_dense20000_p$p.json" k_dense_hess_epi hbm_bytes_per_hessvec --only k_dense_sym,k_sym_fold,k_dense_partial3,k_dense_hess_epi "$OUT/dense20000p${p}_fetch" "$OUT/dense20000p${p}_write"
done
[ -d "$OUT/bqp60_fetch" ] && python3 tools/pmc_sum.py "$OUT/pmc_bqp60_p32.json" k_dense_hess_epi hbm_bytes_per_hessvec --only k_gram_mfma,k_gram_apply,k_adjoint_gram,k_adjoint_tiled,k_dense_partial3,k_dense_sym,k_sym_fold,k_dense_hess_epi "$OUT/bqp60_fetch" "$OUT/bqp60_write"
[ -d "$OUT/theta5000_fetch" ] && python3 tools/pmc_sum.py "$OUT/pmc_theta5000_p32.json" k_sph_hess_fused hbm_bytes_per_hessvec --only k_sddmm,k_support_spmm,k_dense_partial3,k_dense_sym,k_sym_fold,k_sph_hess "$OUT/theta5000_fetch" "$OUT/theta5000_write"
[ -d "$OUT/hess1e6_fetch" ] && python3 tools/pmc_to_json.py k_hess_ "$OUT/pmc_hess_n1e6_p32.json" "$OUT/hess1e6_fetch" "$OUT/hess1e6_write"
ls "$OUT" | grep -v "^[a-z0-9_]*$"
