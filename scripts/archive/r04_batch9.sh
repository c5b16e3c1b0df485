#!/bin/bash
# Round 4, GPU batch 9: where does the per-tile MaxScore lose?  Diagnostics builds (HEAD and working tree) with the timers on: raw
# entries, refinements, phase tables; then the product A/B.
set -o pipefail
mkdir -p gpurun_out
for v in diag_prev diag; do
  for w in "c2:" "k100:--k 100"; do
    DS_LIBRARY=variants/lib_$v.so DS_ALLOW_STALE_LIBRARY=1 DS_PHASE_TIMERS=1 DS_PHASE_DUMP=1 DS_BENCH_SURFACE=0 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 16 ${w#*:} \
      > gpurun_out/r04i_${v}_${w%%:*}.json 2> gpurun_out/r04i_${v}_${w%%:*}.log || { tail -5 gpurun_out/r04i_${v}_${w%%:*}.log; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/r04i_${v}_${w%%:*}.json')); print('$v ${w%%:*}', d['diagnostics']['wave_refines'], d['diagnostics']['raw_entries'], d['diagnostics']['refine_survivors'], d['diagnostics']['raw_entries_sparse'], 'req/q', round(d['roofline']['bytes_per_query']))"
    python scripts/phase_report.py gpurun_out/r04i_${v}_${w%%:*}.log | head -8
  done
done
bash scripts/ab_r04.sh r04i "c2 k100" variants/lib_prev.so 2>&1 | tee gpurun_out/r04i_ab.txt || exit 1
echo R04I_OK
