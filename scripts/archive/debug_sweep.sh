#!/bin/bash
# Timing experiments with parts of the fast kernel switched off (DS_DEBUG bits; results are NOT valid in these runs).
mkdir -p gpurun_out
for dbg in "$@"; do
    DS_DEBUG=$dbg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --check 0 > gpurun_out/dbg_$dbg.json 2> gpurun_out/dbg_$dbg.log || { echo "debug $dbg failed"; tail -3 gpurun_out/dbg_$dbg.log; exit 1; }
    python -c "
import json; d = json.load(open('gpurun_out/dbg_$dbg.json')); s = d['stages_ms']
print('debug %-3s topk %.2f ms dense %.2f ms' % ('$dbg', s['ds_jaccard_topk_kernel'], s['ds_jaccard_dense_kernel']))"
done
