#!/bin/bash
# round 5, first GPU pass of the per-wave tile loop: parity tests of the Jaccard kernels, then an A/B against round 4's library
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_jaccard.py tests/test_gpu_guards.py tests/test_gpu_property.py -x -q -m gpu > gpurun_out/r05_first_tests.log 2>&1
status=$?
tail -5 gpurun_out/r05_first_tests.log
[ $status -ne 0 ] && exit $status
bash scripts/ab_r04.sh r05a "c2 k100 c3s" variants/lib_r04.so
