#!/bin/bash
# narrow against wide geometry on shapes around the crossover (round 5 re-check): fast kernel ms of one launch each
mkdir -p gpurun_out
for shape in "5m:--truth 5000000 --queries 50000 --k 50" "10m:--truth 10000000 --queries 30000 --k 50" "20m:--truth 20000000 --queries 20000 --k 100"; do
  for g in narrow wide; do
    label=${shape%%:*}_$g
    DS_GEOMETRY=$g DS_BENCH_SURFACE=0 timeout -k 10 600 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 16 ${shape#*:} > gpurun_out/geom_$label.json 2> gpurun_out/geom_$label.log || { echo "$label failed"; tail -3 gpurun_out/geom_$label.log; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/geom_$label.json')); s=d['stages_ms']; print('$label', round(d['value']), 'fast %.2f' % s['ds_jaccard_topk_kernel'], 'features %.2f' % s['construct_features'], 'tiles', d['tiles'], 'redos', d['sparse_redos'])"
  done
done
