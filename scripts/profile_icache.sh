#!/bin/bash
# Instruction-cache counters of a bench workload (rocprofv3 --pmc with --kernel-trace only).
# Usage: bash scripts/profile_icache.sh [tag] [bench args]; DS_LIBRARY / DS_ALLOW_STALE_LIBRARY exported by the caller select a variant.
tag=${1:-default}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_icache_${tag} -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 0 "$@" > gpurun_out/pmc_icache_${tag}.json 2> gpurun_out/pmc_icache_${tag}.log
echo exit=$?
python3 - <<PY | tee gpurun_out/pmc_icache_${tag}_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_icache_${tag}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in agg.items():
    if "ds::" in k: print(k, dict(c))
PY
