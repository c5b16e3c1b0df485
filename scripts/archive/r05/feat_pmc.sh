#!/bin/bash
# counters of the features kernel: this build and round 4's library, C2
KERNEL=ds_construct_features_kernel bash scripts/r05/pmc_quick.sh featnew "$@" | grep -E "mean ms|INSTS|wait share"
DS_LIBRARY=variants/lib_r04.so DS_ALLOW_STALE_LIBRARY=1 KERNEL=ds_construct_features_kernel bash scripts/r05/pmc_quick.sh featold "$@" | grep -E "mean ms|INSTS|wait share"
