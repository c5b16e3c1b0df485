#!/bin/bash
# Round 4, GPU batch 5: two-level collect test with the round's atomics batched (tl2) against two-level (tl1) and the working
# library (forward index), four workloads.
set -o pipefail
mkdir -p gpurun_out
bash scripts/ab_r04.sh r04e "c2 k100 c3s c5s" variants/lib_tl1.so variants/lib_tl2.so 2>&1 | tee gpurun_out/r04e_ab.txt || exit 1
echo R04E_OK
