#!/bin/bash
# Per-phase cycle table of ds_jaccard_topk_kernel: diagnostics build (variants/lib_diag.so, -DDS_DIAGNOSTICS, same sources)
# with DS_PHASE_TIMERS=1.  Usage on the GPU box: bash scripts/phase_run.sh <tag> [bench args]
tag=${1:-r02}; shift
mkdir -p gpurun_out
DS_LIBRARY=variants/lib_diag.so DS_ALLOW_STALE_LIBRARY=${DS_ALLOW_STALE_LIBRARY:-0} DS_PHASE_TIMERS=1 DS_PHASE_DUMP=1 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 0 "$@" \
    > gpurun_out/phase_${tag}.json 2> gpurun_out/phase_${tag}.log || { tail -5 gpurun_out/phase_${tag}.log; exit 1; }
python scripts/phase_report.py gpurun_out/phase_${tag}.log | tee gpurun_out/phase_${tag}_table.txt
