#!/bin/bash
# where the fast kernel's instructions go: the diagnostics build with parts of the sparse sweep switched off (DS_DEBUG bits;
# answers are wrong, no check), two SQ counter passes each
for bits in ${BITS:-0 1 4 2}; do
  echo "== DS_DEBUG=$bits"
  DS_LIBRARY=variants/lib_diag.so DS_ALLOW_STALE_LIBRARY=1 DS_DEBUG=$bits bash scripts/r05/pmc_quick.sh split$bits "$@" | grep -E "mean ms|INSTS_VALU|INSTS_SALU|INSTS_LDS|INSTS_VMEM|INSTS_SMEM|wait share"
done
