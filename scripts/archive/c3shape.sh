#!/bin/bash
# C3-shape check on one GPU box: 5M truth titles, a 50k-query sample of C3's queries, top-50 (tie behaviour, dense-path share).
tag=${1:-c3shape}
mkdir -p gpurun_out
timeout -k 10 ${LIMIT:-800} python bench.py --truth 5000000 --queries ${QUERIES:-50000} --k 50 --steps 1 --warmup 1 --cpu-seconds 0 --check ${CHECK:-16} \
    > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.log || { tail -5 gpurun_out/bench_${tag}.log; exit 2; }
python - <<PY
import json
d = json.load(open("gpurun_out/bench_${tag}.json"))
print(round(d["value"]), d["stages_ms"], d["dense_reasons"], d["dense_path_queries"], d["exact_candidates_per_query"])
PY
