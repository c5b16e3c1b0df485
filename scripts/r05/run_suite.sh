#!/bin/bash
# the whole GPU suite (progress into gpurun_out so that a long run is seen alive), then a default bench line
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r05_suite.log 2>&1
status=$?
tail -15 gpurun_out/r05_suite.log
[ $status -ne 0 ] && exit $status
timeout -k 10 600 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.log || { tail -5 gpurun_out/r05_bench_default.log; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_default.json')); print(round(d['value']), d['ms_per_step'], d['stages_ms'], d['roofline']['frac'], d['roofline'].get('limited_by'), d['roofline'].get('memory_level'), d['cpu_baseline']['value'])"
