#!/bin/bash
# round 5: features kernel with truth records + per-query staging, narrow forward index, index options -- the GPU suite without
# the 5M / 50M-row configurations, then C2 / top-100 against round 4's library
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not c3_full and not c5 and not 50m and not scale" > gpurun_out/r05_feat_tests.log 2>&1 || { tail -25 gpurun_out/r05_feat_tests.log; exit 1; }
tail -2 gpurun_out/r05_feat_tests.log
bash scripts/ab_r04.sh r05f "c2 k100" variants/lib_r04.so
