#!/bin/bash
# Parity soak on one box (round 5): the hypothesis property tests with fresh examples, then bench workloads of other seeds / sizes /
# top_n with thousands of queries (rows AND features) verified against the oracle.  A mismatch fails the run (bench.py asserts).
# Usage (gpurun): bash scripts/r05/soak.sh [examples]
examples=${1:-600}
mkdir -p gpurun_out
DS_PROPERTY_EXAMPLES=$examples timeout -k 10 900 python -m pytest tests/test_gpu_property.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_soak_property.log 2>&1 || { tail -30 gpurun_out/r05_soak_property.log; exit 1; }
tail -1 gpurun_out/r05_soak_property.log
run() {  # label, bench args
  label=$1; shift
  DS_BENCH_SURFACE=0 timeout -k 10 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > gpurun_out/r05_soak_${label}.json 2> gpurun_out/r05_soak_${label}.log || { echo "$label FAILED"; tail -5 gpurun_out/r05_soak_${label}.log; exit 2; }
  python -c "
import json; d=json.load(open('gpurun_out/r05_soak_${label}.json')); print('$label', 'verified', d['verified_queries'], 'slow', d['dense_path_queries'], d['dense_reasons'], 'redos', d['sparse_redos'])"
}
for seed in 11 12 13 14; do
  run c2_k10_s$seed --seed $seed --check 6000 || exit 2
  run c2_k100_s$seed --seed $seed --k 100 --check 1500 || exit 2
  run c2_k37_s$seed --seed $seed --k 37 --check 3000 || exit 2
  run small_s$seed --seed $seed --truth 60000 --queries 20000 --k 10 --check 6000 || exit 2
  run mid_s$seed --seed $seed --truth 1500000 --queries 30000 --k 25 --check 2000 || exit 2
done
run c3s_s21 --seed 21 --truth 5000000 --queries 50000 --k 50 --check 600 || exit 2
run c5s_s22 --seed 22 --config C5 --queries 20000 --check 150 || exit 2
echo R05_SOAK_OK
