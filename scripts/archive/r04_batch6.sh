#!/bin/bash
# Round 4, GPU batch 6: GPU suite on the working library (two-level collect test, forward index, split rank counting), the
# bounds-checking build on the bench workloads, then rounds of 4 / 2 items per wave against the default 3.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04f_pytest.log 2>&1 || { tail -40 gpurun_out/r04f_pytest.log; exit 1; }
tail -2 gpurun_out/r04f_pytest.log
bash scripts/bounds_bench.sh 2>&1 | tee gpurun_out/r04f_bounds.txt
grep -q failed gpurun_out/r04f_bounds.txt && exit 1
bash scripts/ab_r04.sh r04f "c2 k100 c3s c5s" variants/lib_r4.so variants/lib_r2.so 2>&1 | tee gpurun_out/r04f_ab.txt || exit 1
echo R04F_OK
