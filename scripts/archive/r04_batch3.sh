#!/bin/bash
# Round 4, GPU batch 3: the whole GPU suite on the working library (epoch halving on redo, 32-bit LCS recurrence in the features
# kernel, staged 9-argument entry), then A/B: prev (round-3 kernel + hand-over fix) / working / two-level collect test (with and
# without the halving), then the phase table at top-100.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04c_pytest.log 2>&1 || { tail -40 gpurun_out/r04c_pytest.log; exit 1; }
tail -2 gpurun_out/r04c_pytest.log
bash scripts/ab_r04.sh r04c "c2 k100 c3s c5s" variants/lib_prev.so variants/lib_twolevel.so variants/lib_twolevel_noshrink.so 2>&1 | tee gpurun_out/r04c_ab.txt || exit 1
DS_ALLOW_STALE_LIBRARY=1 DS_BENCH_SURFACE=0 bash scripts/phase_run.sh r04c_k100 --k 100 --check 16 || exit 1
echo R04C_OK
