set -x
python -m pytest tests/test_gpu_dense.py tests/test_gpu_affine.py -q -x 2>&1 | tail -15
python -m pytest tests/test_gpu_onlyunitdiag.py -q -x -k "windowed" 2>&1 | tail -5
python tools/densesym_probe.py 20000 16 32 2>&1 | tail -12
python tools/densesym_probe.py 5000 32 2>&1 | tail -6
python tools/densesym_probe.py 1840 32 2>&1 | tail -6
python tools/affine_chain_probe.py 2>&1 | tail -24
python tools/hess_large_probe.py 1000 32 2>&1 | tail -4
for s in gpp6 b30x100 b10x200 b30x200s; do timeout 150 python tools/thetaG51_opts.py $s 2>&1 | tail -1; done
