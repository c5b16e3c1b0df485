#!/bin/bash
# Timing experiments with the diagnostics build (results are WRONG by design; only the kernel time is of interest):
# DS_DEBUG bit 1 = no scatter / collect work on sparse tiles, 2 = collect takes but does not test, 16 = collect does not take.
mkdir -p gpurun_out
for bits in ${DS_BITS:-0 1 2 16 64 18 82}; do
  DS_LIBRARY=variants/lib_diag.so DS_DEBUG=$bits timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --check 0 > gpurun_out/debug_bits_$bits.json 2> gpurun_out/debug_bits_$bits.log || { tail -3 gpurun_out/debug_bits_$bits.log; }
  python -c "
import json; d=json.load(open('gpurun_out/debug_bits_$bits.json')); print('DS_DEBUG=$bits', d['stages_ms']['ds_jaccard_topk_kernel'], d['tiles'], d['selections_per_query'], d['dense_path_queries'])"
done
