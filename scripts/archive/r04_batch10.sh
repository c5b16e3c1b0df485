#!/bin/bash
# Round 4, GPU batch 10: adaptive epochs in the large-k instantiation only (top-100, narrow): fill rule (working library) against the
# plain ramp and HEAD; C2 at top-10 as the control (its instantiation does not change).
set -o pipefail
mkdir -p gpurun_out
bash scripts/ab_r04.sh r04j "k100 c2" variants/lib_head.so variants/lib_rampk.so 2>&1 | tee gpurun_out/r04j_ab.txt || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_guards.py tests/test_gpu_jaccard.py tests/test_gpu_property.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
echo R04J_OK
