#!/bin/bash
# Round 4, GPU batch 12: a posting's 16 information bits as 16 membership bits (no sums code) against HEAD: parity tests under the
# variant, then A/B on four workloads.
set -o pipefail
mkdir -p gpurun_out
DS_LIBRARY=variants/lib_sig16.so DS_ALLOW_STALE_LIBRARY=1 timeout -k 10 800 python -m pytest tests/test_gpu_jaccard.py tests/test_gpu_jaccard_classes.py tests/test_gpu_property.py tests/test_gpu_configs.py -x -q -k "not geometr" > gpurun_out/r04l_pytest.log 2>&1 || { tail -30 gpurun_out/r04l_pytest.log; exit 1; }
tail -1 gpurun_out/r04l_pytest.log
bash scripts/ab_r04.sh r04l "c2 k100 c3s c5s" variants/lib_head.so variants/lib_sig16.so 2>&1 | tee gpurun_out/r04l_ab.txt || exit 1
echo R04L_OK
