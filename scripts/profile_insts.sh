#!/bin/bash
# Instruction-mix counters of the default bench workload (rocprofv3 --pmc with --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_insts -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 0 "$@" > gpurun_out/pmc_insts.json 2> gpurun_out/pmc_insts.log
echo exit=$?
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_insts/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, c in agg.items():
    if "topk" in k or "features" in k: print(k, {name: "%.4g" % (v / n[(k, name)]) for name, v in c.items()})
PY
