#!/bin/bash
# Round 4, GPU batch 2: new tests, then A/B of the adaptive epoch length (working library: first epoch 1 tile) against the
# round-3 kernel with the hand-over fix (prev), first epoch 2 tiles (adapt2) and the rolled append paths (rolled: 49 / 53 KB of
# code instead of 81 / 88), then the instruction cache of the working library and of the rolled one.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_surface.py tests/test_gpu_guards.py -x -q > gpurun_out/r04b_pytest.log 2>&1 || { tail -40 gpurun_out/r04b_pytest.log; exit 1; }
tail -2 gpurun_out/r04b_pytest.log
bash scripts/ab_r04.sh r04b "c2 k100 c3s c5s" variants/lib_prev.so variants/lib_adapt2.so variants/lib_rolled.so 2>&1 | tee gpurun_out/r04b_ab.txt || exit 1
DS_BENCH_SURFACE=0 bash scripts/profile_icache.sh r04b_default || exit 1
export DS_LIBRARY=variants/lib_rolled.so DS_ALLOW_STALE_LIBRARY=1 DS_BENCH_SURFACE=0
bash scripts/profile_icache.sh r04b_rolled || exit 1
echo R04B_OK
