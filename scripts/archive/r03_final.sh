#!/bin/bash
# The round's reference numbers on ONE box: smoke, kernel-trace stats of the default bench, PMC passes of C2 and C3, bench
# lines of C2 (with the CPU baseline), top-100, C3 and the C5 shard.  Usage (gpurun): bash scripts/r03_final.sh [tag]
tag=${1:-r03_final}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 || { tail -5 gpurun_out/${tag}_smoke.log; exit 1; }
bash scripts/profile_bench.sh ${tag} > gpurun_out/${tag}_kernel_stats.txt 2>&1 || { tail -5 gpurun_out/${tag}_kernel_stats.txt; exit 2; }
bash scripts/profile_pmc.sh ${tag}_c2 > gpurun_out/${tag}_pmc_c2.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c2.txt; exit 3; }
bash scripts/profile_pmc.sh ${tag}_c3 --config C3 > gpurun_out/${tag}_pmc_c3.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c3.txt; exit 4; }
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.log || { tail -3 gpurun_out/${tag}_c2_bench.log; exit 5; }
timeout -k 10 300 python bench.py --k 100 --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/${tag}_c2_k100_bench.json 2> gpurun_out/${tag}_c2_k100_bench.log || exit 6
timeout -k 10 300 python bench.py --config C3 --steps 2 --warmup 1 --cpu-seconds 10 --check 32 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.log || { tail -3 gpurun_out/${tag}_c3_bench.log; exit 7; }
DS_BUILD_LOG=1 timeout -k 10 400 python bench.py --config C5 --queries 125000 --steps 2 --warmup 1 --cpu-seconds 0 --check 8 > gpurun_out/${tag}_c5shard_bench.json 2> gpurun_out/${tag}_c5shard_bench.log || { tail -3 gpurun_out/${tag}_c5shard_bench.log; exit 8; }
python - <<PY
import json
for name in ("c2", "c2_k100", "c3", "c5shard"):
    d = json.load(open("gpurun_out/${tag}_%s_bench.json" % name))
    print(name, round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], d["dense_reasons"], "roofline %.3f" % d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "x ref floor %.2f" % d["speedup_over_reference_hbm_floor"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
