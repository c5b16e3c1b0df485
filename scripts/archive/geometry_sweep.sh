#!/bin/bash
# Which geometry for which truth-set size: the default library forced wide / narrow, plus variant libraries forced narrow.
mkdir -p gpurun_out
run() {  # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 600 python bench.py --cpu-seconds 0 --check 8 --steps 2 --warmup 1 "$@" > gpurun_out/geo_$label.json 2> gpurun_out/geo_$label.log || { echo "$label failed"; tail -2 gpurun_out/geo_$label.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/geo_$label.json')); print('$label', round(d['value']), d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['dense_reasons']['ties'])"
}
for n in 1200000 2500000; do
  run wide_$n DS_GEOMETRY=wide -- --truth $n --queries 50000 --k 10
  run narrow_$n DS_GEOMETRY=narrow -- --truth $n --queries 50000 --k 10
done
for v in "$@"; do
  name=$(basename $v .so)
  run ${name}_c2 DS_GEOMETRY=narrow DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1 --
  run ${name}_c3s DS_GEOMETRY=narrow DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1 -- --truth 5000000 --queries 50000 --k 50
done
