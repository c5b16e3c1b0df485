#!/bin/bash
# A/B/C... on one box: the working library and any number of variant libraries on three workloads (C2 narrow, top-100,
# C3 shape wide).  Usage: bash scripts/ab_many.sh <tag> <variant.so> [<variant.so> ...]
tag=$1; shift
mkdir -p gpurun_out
run() {  # label, library env, bench args...
  label=$1; lib=$2; shift 2
  env $lib timeout -k 10 600 python bench.py --cpu-seconds 0 --check 16 "$@" > gpurun_out/abm_${tag}_${label}.json 2> gpurun_out/abm_${tag}_${label}.log || { echo "$label failed"; tail -3 gpurun_out/abm_${tag}_${label}.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/abm_${tag}_${label}.json')); print('$label', round(d['value']), d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['verified_queries'])"
}
for w in c2 k100 c3s; do
  case $w in
    c2) args="--steps 3 --warmup 1";;
    k100) args="--k 100 --steps 2 --warmup 1";;
    c3s) args="--truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1";;
  esac
  run ${w}_default "DS_X=0" $args
  for v in "$@"; do
    run ${w}_$(basename $v .so) "DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1" $args
  done
done
