#!/bin/bash
# round 5, first GPU pass: correctness of the EARLY persistent trip + the review fixes, then timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_onlyunitdiag.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -15 > gpurun_out/a_tests_persist.log
timeout 900 python tools/early_probe.py 32,16,8 > gpurun_out/a_early_probe.log 2>&1
timeout 300 python tools/persist_timeline.py 32 1 > gpurun_out/a_timeline_early1.log 2>&1
timeout 300 python tools/persist_timeline.py 32 0 > gpurun_out/a_timeline_early0.log 2>&1
timeout 900 python -m pytest "tests/test_gpu_affine.py::test_sphere_hessvec_with_more_long_constraints_than_waves" "tests/test_gpu_multiblock.py::test_dense_slack_entry_points_refuse_a_blocked_handle" "tests/test_gpu_dense.py::test_dense_operators_at_the_benchmarked_size" -x -q 2>&1 | tail -15 > gpurun_out/a_tests_new.log
tail -5 gpurun_out/a_tests_persist.log; tail -60 gpurun_out/a_early_probe.log; tail -5 gpurun_out/a_tests_new.log
