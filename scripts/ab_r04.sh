#!/bin/bash
# A/B/C... on one box (round 4): the working library and any number of variant libraries on four workloads -- C2 (narrow),
# C2 at the reference's prediction top_n = 100, the C3 shape (narrow, 50k queries x 5M rows, top-50), the C5 shape (wide, 20k
# queries x 50M rows, top-100).  Prints pairs/s, fast kernel ms, literal kernel ms, features ms, epoch redos, verified queries.
# Usage: bash scripts/ab_r04.sh <tag> "<workloads>" <variant.so> [<variant.so> ...]      workloads e.g. "c2 k100 c3s c5s"
tag=$1; workloads=$2; shift 2
mkdir -p gpurun_out
run() {  # label, library env, bench args...
  label=$1; lib=$2; shift 2
  env $lib DS_BENCH_SURFACE=0 timeout -k 10 900 python bench.py --cpu-seconds 0 --check 16 "$@" > gpurun_out/abm_${tag}_${label}.json 2> gpurun_out/abm_${tag}_${label}.log || { echo "$label failed"; tail -3 gpurun_out/abm_${tag}_${label}.log; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/abm_${tag}_${label}.json')); s=d['stages_ms']; print('$label', round(d['value']), 'fast %.3f' % s['ds_jaccard_topk_kernel'], 'literal %.3f' % s['ds_jaccard_dense_kernel'], 'features %.3f' % s['construct_features'], 'redos', d['sparse_redos'], 'slow', d['dense_path_queries'], 'verified', d['verified_queries'], 'tiles', d['tiles'])"
}
for w in $workloads; do
  case $w in
    c2) args="--steps 3 --warmup 1";;
    k100) args="--k 100 --steps 2 --warmup 1";;
    c3s) args="--truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1";;
    c5s) args="--config C5 --queries 20000 --steps 1 --warmup 1";;
  esac
  run ${w}_default "DS_X=0" $args || exit 1
  for v in "$@"; do
    run ${w}_$(basename $v .so) "DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1" $args || exit 1
  done
done
