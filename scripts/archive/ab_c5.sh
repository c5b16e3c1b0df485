#!/bin/bash
# A/B at the C5 shape (50M truth titles, top-100, 20000 queries of one GPU's shard): the working library and variant libraries.
tag=$1; shift
mkdir -p gpurun_out
run() {
  label=$1; lib=$2; shift 2
  env $lib timeout -k 10 500 python bench.py --cpu-seconds 0 --check 8 --config C5 --queries 20000 --steps 1 --warmup 1 > gpurun_out/abc5_${tag}_${label}.json 2> gpurun_out/abc5_${tag}_${label}.log || { echo "$label failed"; tail -3 gpurun_out/abc5_${tag}_${label}.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/abc5_${tag}_${label}.json')); print('$label', round(d['value']), d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['verified_queries'], d['tiles'], d['roofline']['requested_bytes_per_launch'])"
}
run default "DS_X=0"
for v in "$@"; do run $(basename $v .so) "DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1"; done
