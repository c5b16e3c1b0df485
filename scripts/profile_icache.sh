#!/bin/bash
# Instruction-cache counters of the default bench workload (rocprofv3 --pmc with --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_icache -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 0 > gpurun_out/pmc_icache.json 2> gpurun_out/pmc_icache.log
echo exit=$?
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in agg.items():
    if "ds::" in k: print(k, dict(c))
PY
