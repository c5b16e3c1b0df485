#!/bin/bash
# round 5: quick parity pass, phase table of the diagnostics build, A/B against round 4 on C2 and top-100
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_jaccard.py tests/test_gpu_guards.py -x -q -m gpu > gpurun_out/r05_tests.log 2>&1 || { tail -5 gpurun_out/r05_tests.log; exit 1; }
tail -2 gpurun_out/r05_tests.log
DS_ALLOW_STALE_LIBRARY=1 bash scripts/phase_run.sh r05_${1:-x}_c2
grep "sub-tiles of thread" gpurun_out/phase_r05_${1:-x}_c2.log | tail -1
bash scripts/ab_r04.sh r05${1:-x} "${2:-c2 k100}" ${3:-variants/lib_r04.so}
