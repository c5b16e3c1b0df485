#!/bin/bash
# Round 4, GPU batch 8: MaxScore per tile (tile_skip).  GPU suite, bounds-checking build on the bench workloads, A/B against HEAD.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04h_pytest.log 2>&1 || { tail -40 gpurun_out/r04h_pytest.log; exit 1; }
tail -2 gpurun_out/r04h_pytest.log
bash scripts/bounds_bench.sh 2>&1 | tee gpurun_out/r04h_bounds.txt
grep -q failed gpurun_out/r04h_bounds.txt && exit 1
bash scripts/ab_r04.sh r04h "c2 k100 c3s c5s" variants/lib_prev.so 2>&1 | tee gpurun_out/r04h_ab.txt || exit 1
echo R04H_OK
