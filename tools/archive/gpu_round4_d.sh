set -x
timeout 900 python -m pytest tests/test_gpu_local_ranks.py -q 2>&1 | tail -15 | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_dense.py tests/test_gpu_affine.py -q 2>&1 | tail -5 | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_edge_cases.py -q 2>&1 | tail -5 | cut -c1-250
timeout 200 python tools/densesym_probe.py 20000 16 32 2>&1 | tail -18
timeout 100 python tools/persist_timeline.py 32 2>&1 | tail -24
timeout 200 python tools/affine_chain_probe.py theta5000 2>&1 | head -4
timeout 200 python tools/affine_chain_probe.py theta5000 --nofuse 2>&1 | head -3
bash tools/profile_round4.sh affine 2>&1 | tail -12
for t in bqp60_p32 theta5000; do echo "== $t"; head -12 gpurun_out/prof4/${t}_kernel_stats.csv | cut -c1-160; done
