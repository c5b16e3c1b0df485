#!/bin/bash
# Round 4, GPU batch 13: MaxScore per tile taken apart -- the mechanism alone (count computed by the builder / at an epoch's first tile,
# refinement by the entry's own tile, but the selection's skip set applied: same work as HEAD) against the full variant and HEAD.
mkdir -p gpurun_out
bash scripts/ab_r04.sh r04m "c2 k100" variants/lib_head.so variants/lib_tsk_dry.so variants/lib_tsk_full.so 2>&1 | tee gpurun_out/r04m_ab.txt
