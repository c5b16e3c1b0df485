#!/bin/bash
# round 5: bank-aware order of the postings inside a (column, tile, parity) part (DS_POSTING_ORDER=1) against ascending rows
mkdir -p gpurun_out
DS_POSTING_ORDER=1 timeout -k 10 600 python -m pytest tests/test_gpu_jaccard.py tests/test_gpu_guards.py tests/test_gpu_property.py -x -q -m gpu > gpurun_out/r05_bank_tests.log 2>&1 || { tail -25 gpurun_out/r05_bank_tests.log; exit 1; }
tail -2 gpurun_out/r05_bank_tests.log
run() { label=$1; shift
  env "$@" DS_BENCH_SURFACE=0 timeout -k 10 900 python bench.py --cpu-seconds 0 --check 16 $ARGS > gpurun_out/bank_${label}.json 2> gpurun_out/bank_${label}.log || { echo "$label failed"; tail -3 gpurun_out/bank_${label}.log; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bank_${label}.json')); s=d['stages_ms']; print('$label', round(d['value']), 'fast %.3f' % s['ds_jaccard_topk_kernel'], 'features %.3f' % s['construct_features'], 'redos', d['sparse_redos'], 'slow', d['dense_path_queries'], 'verified', d['verified_queries'])"
}
for w in c2 k100 c3s c5s; do
  case $w in
    c2) ARGS="--steps 3 --warmup 1";;
    k100) ARGS="--k 100 --steps 2 --warmup 1";;
    c3s) ARGS="--truth 5000000 --queries 50000 --k 50 --steps 1 --warmup 1";;
    c5s) ARGS="--config C5 --queries 20000 --steps 1 --warmup 1";;
  esac
  run ${w}_ascending DS_POSTING_ORDER=0 || exit 1
  run ${w}_bankdeal DS_POSTING_ORDER=1 || exit 1
done
DS_POSTING_ORDER=0 bash scripts/r05/pmc_quick.sh bank0 | grep -E "mean ms|bank conflict|LDS_IDX|BANK"
DS_POSTING_ORDER=1 bash scripts/r05/pmc_quick.sh bank1 | grep -E "mean ms|bank conflict|LDS_IDX|BANK"
