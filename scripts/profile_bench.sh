#!/bin/bash
# Kernel-trace profile of the default bench workload (run on the GPU box through gpurun; summaries are copied to
# profiles/ by hand afterwards).  Usage: bash scripts/profile_bench.sh <tag> [bench args...]
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --check 0 "$@" > gpurun_out/prof_${tag}.json 2> gpurun_out/prof_${tag}.log
echo exit=$?
find gpurun_out/prof_${tag} -name "*kernel_stats*" -exec cat {} \;
