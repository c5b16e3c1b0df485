#!/bin/bash
# Calibration run on the GPU box: issue rates (plain run) and FETCH_SIZE / WRITE_SIZE factors per access shape (PMC passes).
# The binary is built in the container (hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/calibrate scripts/micro/calibrate.hip).
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 120 scripts/micro/bin/calibrate issue > gpurun_out/calibrate_${tag}_issue.txt 2>&1 || { tail -5 gpurun_out/calibrate_${tag}_issue.txt; exit 1; }
cat gpurun_out/calibrate_${tag}_issue.txt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/calibrate_${tag}_fetch -- scripts/micro/bin/calibrate fetch > gpurun_out/calibrate_${tag}_fetch.txt 2>&1 || { tail -5 gpurun_out/calibrate_${tag}_fetch.txt; exit 2; }
grep "known bytes" gpurun_out/calibrate_${tag}_fetch.txt
python3 - <<PY
import csv, glob, collections
files = glob.glob("gpurun_out/calibrate_${tag}_fetch/**/*counter_collection.csv", recursive=True)
for f in files:
    for row in csv.DictReader(open(f)):
        print(row["Kernel_Name"][:60], row["Counter_Name"], row["Counter_Value"])
PY
