#!/bin/bash
# two SQ counter passes of the fast kernel on one bench configuration, raw per-launch values (round 5 tuning)
# usage: bash scripts/r05/pmc_quick.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift; counters=$1; shift
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d gpurun_out/pmcq_${tag}_${name} -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --check 0 "$@" > gpurun_out/pmcq_${tag}_${name}.json 2> gpurun_out/pmcq_${tag}_${name}.log || { echo "$name failed"; tail -3 gpurun_out/pmcq_${tag}_${name}.log; return 1; }
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "$@" || exit 1
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "$@" || exit 1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); calls = collections.Counter(); dur = []
for name in ("sq1", "sq2"):
    for path in glob.glob("gpurun_out/pmcq_${tag}_%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(path)):
            if "${KERNEL:-ds_jaccard_topk_kernel}" not in row["Kernel_Name"]: continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); calls[row["Counter_Name"]] += 1
    for path in glob.glob("gpurun_out/pmcq_${tag}_%s/**/*kernel_trace.csv" % name, recursive=True):
        for row in csv.DictReader(open(path)):
            if "${KERNEL:-ds_jaccard_topk_kernel}" in row["Kernel_Name"]: dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
n_disp = {c: calls[c] for c in calls}
per = {c: tot[c] / max(1, len(dur) // 2) for c in tot}   # per launch (each pass sees the same launches)
print("launches per pass", len(dur) // 2, "mean ms", sum(dur) / len(dur) / 1e6)
for c in sorted(per): print("%-24s %.4g" % (c, per[c]))
w = per.get("SQ_WAVE_CYCLES", 1)
print("wait share %.3f  LDS-inst-active share %.3f  VALU-active share %.3f  bank conflict / lds active %.3f" % (per["SQ_WAIT_ANY"] / w, per["SQ_ACTIVE_INST_LDS"] / w, per["SQ_ACTIVE_INST_VALU"] / w, per["SQ_LDS_BANK_CONFLICT"] / max(1, per["SQ_LDS_IDX_ACTIVE"])))
PY
