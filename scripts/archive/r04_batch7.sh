#!/bin/bash
# Round 4, GPU batch 7: features kernel with 16 lanes per pair (four pairs per wave) against the default 32: parity tests under each
# variant, then features time on C2 at top-10 and top-100.
set -o pipefail
mkdir -p gpurun_out
for v in feat16 feat16w2; do
  DS_LIBRARY=variants/lib_$v.so DS_ALLOW_STALE_LIBRARY=1 timeout -k 10 600 python -m pytest tests/test_gpu_features.py tests/test_gpu_surface.py tests/test_gpu_pairs.py -x -q > gpurun_out/r04g_pytest_$v.log 2>&1 || { tail -30 gpurun_out/r04g_pytest_$v.log; exit 1; }
  tail -1 gpurun_out/r04g_pytest_$v.log
done
bash scripts/ab_r04.sh r04g "c2 k100 c5s" variants/lib_feat16.so variants/lib_feat16w2.so 2>&1 | tee gpurun_out/r04g_ab.txt || exit 1
echo R04G_OK
