#!/bin/bash
# The bench workloads under the bounds-checking build of the PRODUCT sources (libdoppel_amd_boundscheck.so: every
# data-dependent global index checked, first violation in the bench line's bounds_record): C2, the C3 shape, the C5 shape.
mkdir -p gpurun_out
run() {
  label=$1; shift
  DS_LIBRARY=doppel-speller_amd/libdoppel_amd_boundscheck.so timeout -k 10 500 python bench.py --cpu-seconds 0 --check 16 --steps 1 --warmup 1 "$@" > gpurun_out/bounds_${label}.json 2> gpurun_out/bounds_${label}.log || { echo "$label failed"; tail -3 gpurun_out/bounds_${label}.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/bounds_${label}.json')); print('$label', 'bounds_record', d['bounds_record'], 'verified', d['verified_queries'], d['stages_ms']['ds_jaccard_topk_kernel'])"
}
run c2
run k100 --k 100
run c3s --truth 5000000 --queries 50000 --k 50
run c5s --config C5 --queries 20000
