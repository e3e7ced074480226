set -x
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_affine.py -q -k "b_route" 2>&1 | tail -4
python -m pytest tests/test_gpu_local_ranks.py -q 2>&1 | tail -12
python -m pytest tests/test_gpu_known_answers.py -q -k "thetaG51" 2>&1 | tail -4
for v in "theta5000 --profile" "theta5000 --profile --nofuse" "bqp60 --profile"; do
  tag=$(echo $v | tr -d ' -')
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o x -- python3 tools/affine_chain_probe.py $v > gpurun_out/prof_$tag.log 2>&1
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); echo "== $v"; head -14 $f | cut -c1-150
done
MSDP_UC_POOL=0 python tools/uc_pool_stress.py 700 2>&1 | tail -6
MSDP_UC_POOL=2 python tools/uc_pool_stress.py 700 2>&1 | tail -6
MSDP_UC_POOL=3 python tools/uc_pool_stress.py 700 2>&1 | tail -6
python tools/uc_pool_stress.py 700 2>&1 | tail -3
