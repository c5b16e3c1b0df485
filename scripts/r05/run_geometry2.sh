#!/bin/bash
# second half of the geometry re-check: 7.5M truth titles, and C2 at top-100 (the wide geometry's larger candidate buffer)
mkdir -p gpurun_out
for shape in "7m5:--truth 7500000 --queries 40000 --k 50" "c2k100:--k 100" "3m:--truth 3000000 --queries 60000 --k 50"; do
  for g in narrow wide; do
    label=${shape%%:*}_$g
    DS_GEOMETRY=$g DS_BENCH_SURFACE=0 timeout -k 10 600 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 16 ${shape#*:} > gpurun_out/geom_$label.json 2> gpurun_out/geom_$label.log || { echo "$label failed"; tail -3 gpurun_out/geom_$label.log; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/geom_$label.json')); s=d['stages_ms']; print('$label', round(d['value']), 'fast %.2f' % s['ds_jaccard_topk_kernel'], 'features %.2f' % s['construct_features'], 'tiles', d['tiles'], 'redos', d['sparse_redos'])"
  done
done
