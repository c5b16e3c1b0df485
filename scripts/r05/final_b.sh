#!/bin/bash
# Round 5 reference numbers, part B (one box): PMC passes and bench lines of C3 (1M x 5M, top-50 = one GPU's shard of C4) and of
# the C5 shard (125k queries x 50M truth titles, top-100 + features = one GPU of eight).
tag=${1:-r05_final}
mkdir -p gpurun_out
export DS_BENCH_SURFACE=0
bash scripts/profile_pmc.sh ${tag}_c3 --config C3 > gpurun_out/${tag}_pmc_c3.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c3.txt; exit 3; }
bash scripts/profile_pmc.sh ${tag}_c5 --config C5 --queries 125000 > gpurun_out/${tag}_pmc_c5.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c5.txt; exit 4; }
unset DS_BENCH_SURFACE
timeout -k 10 300 python bench.py --config C3 --steps 2 --warmup 1 --cpu-seconds 10 --check 32 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.log || { tail -3 gpurun_out/${tag}_c3_bench.log; exit 7; }
DS_BUILD_LOG=1 timeout -k 10 400 python bench.py --config C5 --queries 125000 --steps 2 --warmup 1 --cpu-seconds 0 --check 8 > gpurun_out/${tag}_c5shard_bench.json 2> gpurun_out/${tag}_c5shard_bench.log || { tail -3 gpurun_out/${tag}_c5shard_bench.log; exit 8; }
python - <<PY
import json
for name in ("c3", "c5shard"):
    d = json.load(open("gpurun_out/${tag}_%s_bench.json" % name))
    print(name, round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], d["dense_reasons"], "roofline %.3f" % d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "x ref floor %.2f" % d["speedup_over_reference_hbm_floor"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), "surface", (d.get("surface") or {}).get("pairs_per_s"))
PY
echo R05_FINAL_B_OK
