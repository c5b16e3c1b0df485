#!/bin/bash
# Round 4, after the fix of the map-ahead hand-over word (one word per directory copy): ONE run each of the GPU suite, the
# bench workloads under the bounds-checking build, and the diagnostics builds with the phase timers on (the binary that
# faulted in round 3).  Steps are chained: a failure stops the script.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_pytest.log 2>&1 || { tail -30 gpurun_out/r04a_pytest.log; exit 1; }
tail -3 gpurun_out/r04a_pytest.log
bash scripts/bounds_bench.sh 2>&1 | tee gpurun_out/r04a_bounds.txt
grep -q failed gpurun_out/r04a_bounds.txt && exit 1
# diagnostics + bounds check + timers: the build that recorded site 3 in round 3
DS_LIBRARY=variants/lib_diag_bc.so DS_ALLOW_STALE_LIBRARY=1 DS_PHASE_TIMERS=1 DS_PHASE_DUMP=1 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 16 \
    > gpurun_out/r04a_diag_bc_c2.json 2> gpurun_out/r04a_diag_bc_c2.log || { tail -5 gpurun_out/r04a_diag_bc_c2.log; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04a_diag_bc_c2.json')); print('diag_bc c2 bounds_record', d['bounds_record'], 'verified', d['verified_queries'], d['stages_ms']['ds_jaccard_topk_kernel'])"
# diagnostics + timers: the build that faulted
DS_ALLOW_STALE_LIBRARY=1 bash scripts/phase_run.sh r04a_c2 --check 16 || exit 1
DS_ALLOW_STALE_LIBRARY=1 bash scripts/phase_run.sh r04a_c3s --truth 5000000 --queries 50000 --k 50 --check 16 || exit 1
DS_ALLOW_STALE_LIBRARY=1 bash scripts/phase_run.sh r04a_c5s --config C5 --queries 20000 --check 16 || exit 1
echo R04A_ALL_OK
