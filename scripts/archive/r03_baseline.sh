#!/bin/bash
# Round-3 numbers on one GPU box: PMC passes of C2 (narrow) and C3 (wide), then the bench lines of C2, C3 and C5's shard.
tag=${1:-r03a}
mkdir -p gpurun_out
bash scripts/profile_pmc.sh ${tag}_c2 > gpurun_out/${tag}_pmc_c2.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c2.txt; exit 1; }
bash scripts/profile_pmc.sh ${tag}_c3 --config C3 > gpurun_out/${tag}_pmc_c3.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c3.txt; exit 2; }
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.log || { tail -3 gpurun_out/${tag}_c2_bench.log; exit 3; }
timeout -k 10 300 python bench.py --config C3 --steps 2 --warmup 1 --cpu-seconds 10 --check 32 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.log || { tail -3 gpurun_out/${tag}_c3_bench.log; exit 4; }
timeout -k 10 400 python bench.py --config C5 --queries 125000 --steps 2 --warmup 1 --cpu-seconds 0 --check 8 > gpurun_out/${tag}_c5shard_bench.json 2> gpurun_out/${tag}_c5shard_bench.log || { tail -3 gpurun_out/${tag}_c5shard_bench.log; exit 5; }
python - <<PY
import json
for name in ("c2", "c3", "c5shard"):
    d = json.load(open("gpurun_out/${tag}_%s_bench.json" % name))
    print(name, round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], d["dense_reasons"], "roofline %.3f" % d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "x ref floor %.2f" % d["speedup_over_reference_hbm_floor"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
