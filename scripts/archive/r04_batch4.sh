#!/bin/bash
# Round 4, GPU batch 4: forward index in the exact stage.  Whole GPU suite (with the two-rank C5 rehearsal at 50M rows), the
# bounds-checking build on the bench workloads, A/B against prev, then the default bench line with the surface record.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s --durations=8 > gpurun_out/r04d_pytest.log 2>&1 || { tail -40 gpurun_out/r04d_pytest.log; exit 1; }
grep -E "rehearsal|passed|failed" gpurun_out/r04d_pytest.log | tail -5
bash scripts/bounds_bench.sh 2>&1 | tee gpurun_out/r04d_bounds.txt
grep -q failed gpurun_out/r04d_bounds.txt && exit 1
bash scripts/ab_r04.sh r04d "c2 k100 c3s c5s" variants/lib_prev.so 2>&1 | tee gpurun_out/r04d_ab.txt || exit 1
timeout -k 10 600 python bench.py --cpu-seconds 5 > gpurun_out/r04d_c2_bench.json 2> gpurun_out/r04d_c2_bench.log || { tail -5 gpurun_out/r04d_c2_bench.log; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04d_c2_bench.json')); print('C2', round(d['value']), d['stages_ms']); print('surface', d['surface'])"
echo R04D_OK
