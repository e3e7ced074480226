#!/bin/bash
# multiblock timings under the per-block storage (default from 16 blocks on) against the embedding (MSDP_MULTIBLOCK_BLOCKED=0)
mkdir -p gpurun_out
L=gpurun_out/p_multiblock_times.log; : > $L
cd examples
for mode in 1 0; do
  echo "== MSDP_MULTIBLOCK_BLOCKED=$mode: example_bqp_sparse 20 x 20" >> ../$L
  MSDP_MULTIBLOCK_BLOCKED=$mode timeout 600 python example_bqp_sparse.py 20 20 >> ../$L 2>&1
  MSDP_MULTIBLOCK_BLOCKED=$mode timeout 600 python example_bqp_sparse.py 20 20 >> ../$L 2>&1
done
echo "== blocked: 100 cliques of 20" >> ../$L
timeout 900 python example_bqp_sparse.py 100 20 >> ../$L 2>&1
echo "== blocked: 400 cliques of 10" >> ../$L
timeout 900 python example_bqp_sparse.py 400 10 >> ../$L 2>&1
cd ..
timeout 600 python tools/multiblock_breakdown.py 100 20 2>&1 | head -8 >> $L
cat $L
