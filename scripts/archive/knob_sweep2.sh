#!/bin/bash
# Selection schedule: first / minimum candidate count that triggers a selection = max(DS_SELECT_K * k, DS_SELECT_MIN).
mkdir -p gpurun_out
run() {  # label, bench args (quoted), env...
  label=$1; args=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --check 16 $args > gpurun_out/knob_$label.json 2> gpurun_out/knob_$label.log || { echo "$label failed"; return; }
  python -c "
import json; d=json.load(open('gpurun_out/knob_$label.json')); print('$label', d['stages_ms']['ds_jaccard_topk_kernel'], d['stages_ms']['ds_jaccard_dense_kernel'], d['selections_per_query'], d['verified_queries'])"
}
run base "" DS_X=0
run k4m32 "" DS_SELECT_MIN=32
run k3m16 "" DS_SELECT_K=3 DS_SELECT_MIN=16
run k2m16 "" DS_SELECT_K=2 DS_SELECT_MIN=16
run k2m32 "" DS_SELECT_K=2 DS_SELECT_MIN=32
run k100_base "--k 100" DS_X=0
run k100_k3 "--k 100" DS_SELECT_K=3
run k100_k2 "--k 100" DS_SELECT_K=2
