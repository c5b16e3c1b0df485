#!/bin/bash
# selection knobs re-swept on the diagnostics build (its launch reads them from the environment; no phase timers): fast kernel ms
mkdir -p gpurun_out
run() { label=$1; envs=$2; shift 2
  env $envs DS_LIBRARY=variants/lib_diag.so DS_ALLOW_STALE_LIBRARY=1 DS_BENCH_SURFACE=0 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --check 16 "$@" > gpurun_out/sel_$label.json 2> gpurun_out/sel_$label.log || { echo "$label failed"; tail -3 gpurun_out/sel_$label.log; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/sel_$label.json')); s=d['stages_ms']; print('$label', 'fast %.3f' % s['ds_jaccard_topk_kernel'], 'selections/query %.2f' % d['selections_per_query'], 'redos', d['sparse_redos'], 'slow', d['dense_path_queries'])"
}
for w in "c2:" "k100:--k 100" "c3s:--truth 5000000 --queries 50000 --k 50"; do
  name=${w%%:*}; args=${w#*:}
  run ${name}_base "DS_X=0" $args || exit 1
  for knobs in "DS_SELECT_K=2" "DS_SELECT_K=4" "DS_SELECT_K=6" "DS_SELECT_MIN=32" "DS_SELECT_GROWTH=3" "DS_SELECT_GROWTH=6" "DS_SELECT_GROWTH=8" "DS_SPARSE_QUADS=2048" "DS_SPARSE_QUADS=8192"; do
    run ${name}_${knobs//=/} "$knobs" $args || exit 1
  done
done
