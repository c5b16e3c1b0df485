#!/bin/bash
# One GPU-box pass: parity tests, the default bench and the k=100 bench; every step bounded, stops at the first failure.
tag=${1:-check}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_${tag}.log 2>&1 || { tail -30 gpurun_out/pytest_${tag}.log; exit 1; }
tail -1 gpurun_out/pytest_${tag}.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds ${CPU_SECONDS:-0} --check 64 > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.log || { tail -5 gpurun_out/bench_${tag}.log; exit 2; }
timeout -k 10 300 python bench.py --k 100 --steps 2 --warmup 1 --cpu-seconds 0 --check 64 > gpurun_out/bench_${tag}_k100.json 2> gpurun_out/bench_${tag}_k100.log || { tail -5 gpurun_out/bench_${tag}_k100.log; exit 3; }
python - <<PY
import json
for name in ("bench_${tag}.json", "bench_${tag}_k100.json"):
    d = json.load(open("gpurun_out/" + name))
    print(name, round(d["value"]), d["stages_ms"], d["dense_reasons"], d["exact_candidates_per_query"], d["selections_per_query"], d["verified_queries"])
PY
