#!/bin/bash
# Runtime-knob sweep on the GPU box: scripts/knob_sweep.sh VAR v1 v2 ...
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
    env $var=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --check 16 > gpurun_out/knob_${var}_$v.json 2> gpurun_out/knob_${var}_$v.log || { echo "$var=$v failed"; tail -3 gpurun_out/knob_${var}_$v.log; exit 1; }
    python -c "
import json; d = json.load(open('gpurun_out/knob_${var}_$v.json')); s = d['stages_ms']
print('$var=%-6s topk %.2f ms dense %.2f ms  tiles %s' % ('$v', s['ds_jaccard_topk_kernel'], s['ds_jaccard_dense_kernel'], d['tiles']))"
done
