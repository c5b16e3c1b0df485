#!/bin/bash
# For round 6, measured on the round's last build (nothing adopted here): wide epochs of 32 tiles at the C5 shape, and the
# narrow / wide crossover again now that the wide geometry runs epochs of 16 (5M and 10M truth rows, top-50, both geometries).
# Usage (gpurun): bash scripts/r05/run_wide_next.sh      (variants/lib_wide_e32.so = these sources with -DDS_WIDE_EPOCH=32)
mkdir -p gpurun_out
run() {  # label, env assignment, bench args...
  label=$1; lib=$2; shift 2
  env $lib DS_BENCH_SURFACE=0 timeout -k 10 400 python bench.py --cpu-seconds 0 --check 16 "$@" > gpurun_out/wn_${label}.json 2> gpurun_out/wn_${label}.log || { echo "$label failed"; tail -3 gpurun_out/wn_${label}.log; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/wn_${label}.json')); s=d['stages_ms']; print('$label', d['roofline'].get('geometry'), round(d['value']), 'fast %.3f' % s['ds_jaccard_topk_kernel'], 'literal %.3f' % s['ds_jaccard_dense_kernel'], 'redos', d['sparse_redos'], 'slow', d['dense_path_queries'], 'verified', d['verified_queries'])"
}
c5s="--config C5 --queries 20000 --steps 1 --warmup 1"
run c5s_e16 "DS_X=0" $c5s || exit 1
run c5s_e32 "DS_LIBRARY=variants/lib_wide_e32.so DS_ALLOW_STALE_LIBRARY=1" $c5s || exit 1
run c5s_e16_again "DS_X=0" $c5s || exit 1
for truth in 5000000 10000000; do
  shape="--truth $truth --queries 50000 --k 50 --steps 1 --warmup 1"
  run t${truth}_narrow "DS_GEOMETRY=narrow" $shape || exit 1
  run t${truth}_wide "DS_GEOMETRY=wide" $shape || exit 1
done
echo R05_WIDE_NEXT_OK
