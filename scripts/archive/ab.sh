#!/bin/bash
# A/B on ONE GPU box (devices differ by several percent): the default library against variant libraries, same workload,
# alternating runs.  Usage: bash scripts/ab.sh <tag> <variant.so> [<variant2.so> ...] -- [bench args]
tag=$1; shift
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for round in 1 2; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --check 0 "$@" > gpurun_out/ab_${tag}_default_${round}.json 2> gpurun_out/ab_${tag}_default_${round}.log || exit 1
  for v in "${variants[@]}"; do
    name=$(basename $v .so)
    DS_LIBRARY=$v DS_ALLOW_STALE_LIBRARY=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --check 0 "$@" > gpurun_out/ab_${tag}_${name}_${round}.json 2> gpurun_out/ab_${tag}_${name}_${round}.log || exit 2
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_${tag}_*.json")):
    d = json.load(open(f)); print(f.split("ab_${tag}_")[1], round(d["value"]), d["stages_ms"]["ds_jaccard_topk_kernel"], d["stages_ms"]["ds_jaccard_dense_kernel"])
PY
