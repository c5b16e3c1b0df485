#!/bin/bash
# A/B over several workloads on one box: default library against one variant.  Usage: bash scripts/ab_multi.sh <tag> <variant.so>
tag=$1; v=$2
mkdir -p gpurun_out
run() {  # label, library env, bench args...
  label=$1; lib=$2; shift 2
  env $lib timeout -k 10 600 python bench.py --cpu-seconds 0 --check 16 "$@" > gpurun_out/abm_${tag}_${label}.json 2> gpurun_out/abm_${tag}_${label}.log || { echo "$label failed"; tail -3 gpurun_out/abm_${tag}_${label}.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/abm_${tag}_${label}.json')); print('$label', round(d['value']), d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['dense_reasons'], d['verified_queries'])"
}
V="DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1"
D="DS_X=0"
run c2_default "$D" --steps 3 --warmup 1
run c2_variant "$V" --steps 3 --warmup 1
run k100_default "$D" --k 100 --steps 2 --warmup 1
run k100_variant "$V" --k 100 --steps 2 --warmup 1
run c3s_default "$D" --truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1
run c3s_variant "$V" --truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1
