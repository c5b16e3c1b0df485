#!/bin/bash
# Parity soak of the round's LAST sources (the wide geometry's epochs went from 8 to 16 tiles after scripts/r05/soak.sh ran): the
# C5 shard's bench line with this build's counters, the property tests with fresh examples (both geometries), then bench workloads
# of other seeds and sizes with the WIDE geometry forced at index creation (DS_GEOMETRY=wide) and at its own sizes (50M truth rows),
# rows AND features verified against the oracle.  A mismatch fails the run.  Usage (gpurun): bash scripts/r05/soak_wide.sh [examples] [seconds]
examples=${1:-300}
limit=${2:-600}
start=$SECONDS
mkdir -p gpurun_out
DS_BENCH_SURFACE=0 timeout -k 10 400 python bench.py --config C5 --queries 125000 --steps 2 --warmup 1 --cpu-seconds 0 --check 8 > gpurun_out/r05_head_c5shard_bench.json 2> gpurun_out/r05_head_c5shard_bench.log || { tail -3 gpurun_out/r05_head_c5shard_bench.log; exit 8; }
echo "C5 shard bench line at $((SECONDS - start)) s"
DS_PROPERTY_EXAMPLES=$examples timeout -k 10 600 python -m pytest tests/test_gpu_property.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_soak_wide_property.log 2>&1 || { tail -30 gpurun_out/r05_soak_wide_property.log; exit 1; }
tail -1 gpurun_out/r05_soak_wide_property.log
run() {  # label, bench args
  [ $((SECONDS - start)) -lt $limit ] || { echo "$1 skipped (time)"; return 0; }
  label=$1; shift
  DS_BENCH_SURFACE=0 timeout -k 10 400 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > gpurun_out/r05_soak_wide_${label}.json 2> gpurun_out/r05_soak_wide_${label}.log || { echo "$label FAILED"; tail -5 gpurun_out/r05_soak_wide_${label}.log; exit 2; }
  python -c "
import json; d=json.load(open('gpurun_out/r05_soak_wide_${label}.json')); print('$label', d['roofline'].get('geometry'), 'verified', d['verified_queries'], 'slow', d['dense_path_queries'], d['dense_reasons'], 'redos', d['sparse_redos'])"
}
run c5s_s31 --seed 31 --config C5 --queries 20000 --check 150 || exit 2
run c5s_s32 --seed 32 --config C5 --queries 20000 --k 25 --check 150 || exit 2
export DS_GEOMETRY=wide
for seed in 41 42; do
  run c2_k10_s$seed --seed $seed --check 4000 || exit 2
  run c2_k100_s$seed --seed $seed --k 100 --check 1000 || exit 2
  run c2_k37_s$seed --seed $seed --k 37 --check 2000 || exit 2
  run small_s$seed --seed $seed --truth 60000 --queries 20000 --k 10 --check 4000 || exit 2
  run mid_s$seed --seed $seed --truth 1500000 --queries 30000 --k 25 --check 1500 || exit 2
done
run c3s_s43 --seed 43 --truth 5000000 --queries 50000 --k 50 --check 400 || exit 2
echo "done at $((SECONDS - start)) s"
echo R05_SOAK_WIDE_OK
