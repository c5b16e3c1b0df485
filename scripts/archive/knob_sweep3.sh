#!/bin/bash
# Run-time knobs re-swept on the round-3 kernel (one box): geometry at the C3 shape and at 2M rows, selection schedule.
mkdir -p gpurun_out
run() {  # label, env, bench args
  label=$1; envs=$2; shift 2
  env $envs timeout -k 10 600 python bench.py --cpu-seconds 0 --check 16 "$@" > gpurun_out/knob3_${label}.json 2> gpurun_out/knob3_${label}.log || { echo "$label failed"; tail -2 gpurun_out/knob3_${label}.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/knob3_${label}.json')); print('$label', round(d['value']), d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['dense_reasons']['ties'], d['dense_reasons']['overflow_sparse'], d['selections_per_query'])"
}
C2="--steps 3 --warmup 1"
C3S="--truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1"
M2="--truth 2000000 --queries 50000 --k 10 --steps 2 --warmup 1"
run c2_base "DS_X=0" $C2
run c2_selk2 "DS_SELECT_K=2" $C2
run c2_selk4 "DS_SELECT_K=4" $C2
run c2_grow3 "DS_SELECT_GROWTH=3" $C2
run c2_grow6 "DS_SELECT_GROWTH=6" $C2
run c2_sq2048 "DS_SPARSE_QUADS=2048" $C2
run c2_sq8192 "DS_SPARSE_QUADS=8192" $C2
run c2_wide "DS_GEOMETRY=wide" $C2
run c3s_base "DS_X=0" $C3S
run c3s_narrow "DS_GEOMETRY=narrow" $C3S
run c3s_selk2 "DS_SELECT_K=2" $C3S
run c3s_grow6 "DS_SELECT_GROWTH=6" $C3S
run m2_narrow "DS_GEOMETRY=narrow" $M2
run m2_wide "DS_GEOMETRY=wide" $M2
