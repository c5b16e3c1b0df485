#!/bin/bash
# top-100: retries reset when an epoch is halved (variant) against the working library, three runs each (the hand-overs vary from run to run)
mkdir -p gpurun_out
for i in 1 2 3; do
  bash scripts/ab_r04.sh r04k$i "k100" variants/lib_reset.so 2>&1
done | tee gpurun_out/r04k_ab.txt
