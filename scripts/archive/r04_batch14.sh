#!/bin/bash
# Round 4, GPU batch 14: exact stage's membership steps four at a time (loads batched) against HEAD; parity tests under the variant.
set -o pipefail
mkdir -p gpurun_out
DS_LIBRARY=variants/lib_exact4.so DS_ALLOW_STALE_LIBRARY=1 timeout -k 10 800 python -m pytest tests/test_gpu_jaccard.py tests/test_gpu_jaccard_classes.py tests/test_gpu_property.py tests/test_gpu_configs.py tests/test_gpu_guards.py -x -q -k "not geometr and not bounds_checking" > gpurun_out/r04n_pytest.log 2>&1 || { tail -30 gpurun_out/r04n_pytest.log; exit 1; }
tail -1 gpurun_out/r04n_pytest.log
bash scripts/ab_r04.sh r04n "c2 k100 c3s c5s" variants/lib_head.so variants/lib_exact4.so 2>&1 | tee gpurun_out/r04n_ab.txt || exit 1
echo R04N_OK
