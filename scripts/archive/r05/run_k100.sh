#!/bin/bash
# top-100 on C2: this build against variants (candidate buffer / workgroups per CU) and round 4's library
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_jaccard.py tests/test_gpu_surface.py -x -q -m gpu > gpurun_out/r05_k100_tests.log 2>&1 || { tail -25 gpurun_out/r05_k100_tests.log; exit 1; }
tail -1 gpurun_out/r05_k100_tests.log
bash scripts/ab_r04.sh r05k "k100" "$@"
