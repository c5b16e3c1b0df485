#!/bin/bash
# Run-time knobs of the fast kernel on the default workload (no rebuild): selection schedule and the sparse-tile limit.
mkdir -p gpurun_out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --check 0 > gpurun_out/knob_$label.json 2> gpurun_out/knob_$label.log || { echo "$label failed"; return; }
  python -c "
import json; d=json.load(open('gpurun_out/knob_$label.json')); print('$label', d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['selections_per_query'], d['tiles'])"
}
run base DS_X=0
run min32 DS_SELECT_MIN=32
run min128 DS_SELECT_MIN=128
run min256 DS_SELECT_MIN=256
run grow3 DS_SELECT_GROWTH=3
run grow6 DS_SELECT_GROWTH=6
run grow8 DS_SELECT_GROWTH=8
run sq2048 DS_SPARSE_QUADS=2048
run sq8192 DS_SPARSE_QUADS=8192
run sq16384 DS_SPARSE_QUADS=16384
