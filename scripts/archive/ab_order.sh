#!/bin/bash
# Experiment: the order in which a launch takes its queries (DS_EXPERIMENT_QUERY_ORDER of bench.py).  Usage: bash scripts/ab_order.sh <tag>
tag=$1
mkdir -p gpurun_out
run() {
  label=$1; order=$2; shift 2
  DS_EXPERIMENT_QUERY_ORDER=$order timeout -k 10 400 python bench.py --cpu-seconds 0 --check 16 "$@" > gpurun_out/abo_${tag}_${label}.json 2> gpurun_out/abo_${tag}_${label}.log || { echo "$label failed"; tail -3 gpurun_out/abo_${tag}_${label}.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/abo_${tag}_${label}.json')); print('$label', round(d['value']), d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['verified_queries'])"
}
for order in "" maxint_asc maxint_desc columns_desc; do
  run c2_${order:-asis} "$order" --steps 3 --warmup 1
  run c3s_${order:-asis} "$order" --truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1
done
