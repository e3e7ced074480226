#!/bin/bash
# PMC passes for the dense S*U kernel (run from the repo root on the GPU box): clock, MFMA busy, LDS conflicts.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/dpmc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d "$OUT/a" -- python3 "$ROOT/tools/dense_pmc_probe.py" 32 64 > "$OUT/a.log" 2>&1; echo "a rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/b" -- python3 "$ROOT/tools/dense_pmc_probe.py" 32 64 > "$OUT/b.log" 2>&1; echo "b rc=$?"
cd "$ROOT"
python3 tools/dense_pmc_summary.py
