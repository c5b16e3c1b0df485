#!/bin/bash
# A/B of this build against variant libraries: bash scripts/r05/run_ab.sh "<workloads>" <variant.so> ...
w=$1; shift
bash scripts/ab_r04.sh r05x "$w" "$@"
