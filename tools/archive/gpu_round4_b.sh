set -x
python -m pytest tests/test_gpu_dense.py tests/test_gpu_affine.py -q 2>&1 | tail -15
python -m pytest tests/test_gpu_local_ranks.py -q -x -k "cross_rank or one_exchange or onlyunitdiag_ranks" 2>&1 | tail -15
python -m pytest tests/test_gpu_edge_cases.py -q -x -k "churn" 2>&1 | tail -8
python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_derivatives.py tests/test_gpu_onlyunitdiag.py -q 2>&1 | tail -8
python tools/densesym_probe.py 20000 16 32 2>&1 | tail -12
python tools/densesym_probe.py 10000 32 2>&1 | tail -6
python tools/affine_chain_probe.py theta5000 2>&1 | tail -10
python tools/persist_timeline.py 32 2>&1 | tail -30
python tools/uc_pool_stress.py 200 2>&1 | tail -4
MSDP_UC_POOL=0 python tools/uc_pool_stress.py 200 2>&1 | tail -6
for s in gpp_al60 gpp_al120; do timeout 150 python tools/thetaG51_opts.py $s 2>&1 | tail -1; done
