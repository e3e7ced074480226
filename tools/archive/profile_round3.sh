#!/bin/bash
# Round-3 rocprofv3 evidence, collected on the GPU box from the repo root (results under gpurun_out/prof3, the summaries are
# then copied into profiles/ by hand).  Every profiled command is `python3 ...` itself under `timeout`; counters are collected
# in their own passes (--kernel-trace + --pmc only); graphs are off (rocprofv3 7.2 crashes on graph replay).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof3
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
stats g81_kkt 300 python3 "$ROOT/tools/g81_escape_profile.py"
pmc g81_kkt_fetch 300 FETCH_SIZE python3 "$ROOT/tools/g81_escape_profile.py"
pmc g81_kkt_write 300 WRITE_SIZE python3 "$ROOT/tools/g81_escape_profile.py"
stats dense20000 300 python3 "$ROOT/tools/dense_probe.py" 20000 16 32 64
pmc dense20000_fetch 300 FETCH_SIZE python3 "$ROOT/tools/dense_probe.py" 20000 32
pmc dense20000_write 300 WRITE_SIZE python3 "$ROOT/tools/dense_probe.py" 20000 32
stats k5shard 300 python3 "$ROOT/tools/dense_probe.py" 100000 64 --shard 8
pmc k5shard_fetch 300 FETCH_SIZE python3 "$ROOT/tools/dense_probe.py" 100000 64 --shard 8
pmc k5shard_write 300 WRITE_SIZE python3 "$ROOT/tools/dense_probe.py" 100000 64 --shard 8
stats bqp60_p32 300 python3 "$ROOT/tools/gram_probe.py" 32
pmc bqp60_fetch 300 FETCH_SIZE python3 "$ROOT/tools/gram_probe.py" 32
pmc bqp60_write 300 WRITE_SIZE python3 "$ROOT/tools/gram_probe.py" 32
stats theta5000 300 python3 "$ROOT/tools/theta_probe.py" 32
pmc theta5000_fetch 300 FETCH_SIZE python3 "$ROOT/tools/theta_probe.py" 32
pmc theta5000_write 300 WRITE_SIZE python3 "$ROOT/tools/theta_probe.py" 32
stats chunked1e6 300 python3 "$ROOT/tools/pmc_probe_large.py" 1000 1000 32
pmc chunked1e6_fetch 300 FETCH_SIZE python3 "$ROOT/tools/pmc_probe_large.py" 1000 1000 32
pmc chunked1e6_write 300 WRITE_SIZE python3 "$ROOT/tools/pmc_probe_large.py" 1000 1000 32
cd "$ROOT"
python3 tools/pmc_to_json.py k_hess_ "$OUT/pmc_hess_g81_p32.json" "$OUT/fetch" "$OUT/write"
python3 tools/pmc_to_json.py k_tcg_persist_obl "$OUT/pmc_persist_g81_p32.json" --per 64 "$OUT/fetch" "$OUT/write"
python3 tools/pmc_to_json.py k_be_step "$OUT/pmc_be_step_g81.json" "$OUT/g81_kkt_fetch" "$OUT/g81_kkt_write"
python3 tools/pmc_to_json.py k_dense_partial3 "$OUT/pmc_dense20000_p32.json" "$OUT/dense20000_fetch" "$OUT/dense20000_write"
python3 tools/pmc_to_json.py k_dense_partial3 "$OUT/pmc_k5shard_p64.json" "$OUT/k5shard_fetch" "$OUT/k5shard_write"
python3 tools/pmc_sum.py "$OUT/pmc_bqp60_p32.json" k_dense_hess_epi hbm_bytes_per_hessvec --only k_gram_mfma,k_gram_apply,k_adjoint_tiled,k_dense_partial3,k_dense_hess_epi "$OUT/bqp60_fetch" "$OUT/bqp60_write"
python3 tools/pmc_sum.py "$OUT/pmc_theta5000_p32.json" k_sph_hess_finish hbm_bytes_per_hessvec --only k_sddmm,k_support_spmm,k_dense_partial3,k_sph_hess_raw,k_sph_hess_finish "$OUT/theta5000_fetch" "$OUT/theta5000_write"
# (the chunked path's default trip is the linear-product one of msdp_trip1.hip; profiles/r3_pmc_chunked_n1e6_p32.json -- the two-launch trip
#  of msdp_trip2.hip -- was collected the same way before it became the default, i.e. with option trip1 = 0)
python3 tools/pmc_sum.py "$OUT/pmc_linear_n1e6_p32.json" k_tcg1_upd hbm_bytes_per_trip --only k_tcg1_upd,k_tcg1_head "$OUT/chunked1e6_fetch" "$OUT/chunked1e6_write"
ls "$OUT" | grep -v "^[a-z0-9_]*$"
