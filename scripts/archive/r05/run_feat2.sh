#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_features.py tests/test_gpu_surface.py tests/test_gpu_property.py tests/test_reference_vectors.py tests/test_gpu_pairs.py -x -q -m gpu > gpurun_out/r05_feat_tests.log 2>&1 || { tail -25 gpurun_out/r05_feat_tests.log; exit 1; }
tail -2 gpurun_out/r05_feat_tests.log
bash scripts/ab_r04.sh r05g "c2 k100" "$@"
