#!/bin/bash
# Round 5 reference numbers, part A (one box): smoke, kernel-trace stats of the default bench, PMC passes of C2 at top-10 and at
# the reference's prediction top_n = 100, bench lines of both (C2 with the CPU baseline and the surface record), phase tables of
# the diagnostics build.  Usage (gpurun): bash scripts/r05_final_a.sh [tag]
tag=${1:-r05_final}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 || { tail -5 gpurun_out/${tag}_smoke.log; exit 1; }
DS_BENCH_SURFACE=0 bash scripts/profile_bench.sh ${tag} > gpurun_out/${tag}_kernel_stats.txt 2>&1 || { tail -5 gpurun_out/${tag}_kernel_stats.txt; exit 2; }
export DS_BENCH_SURFACE=0
bash scripts/profile_pmc.sh ${tag}_c2 > gpurun_out/${tag}_pmc_c2.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c2.txt; exit 3; }
bash scripts/profile_pmc.sh ${tag}_k100 --k 100 > gpurun_out/${tag}_pmc_k100.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_k100.txt; exit 4; }
unset DS_BENCH_SURFACE
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.log || { tail -3 gpurun_out/${tag}_c2_bench.log; exit 5; }
timeout -k 10 300 python bench.py --k 100 --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/${tag}_c2_k100_bench.json 2> gpurun_out/${tag}_c2_k100_bench.log || exit 6
for w in "c2:" "k100:--k 100" "c3s:--truth 5000000 --queries 50000 --k 50" "c5s:--config C5 --queries 20000"; do
  DS_ALLOW_STALE_LIBRARY=1 DS_BENCH_SURFACE=0 bash scripts/phase_run.sh ${tag}_${w%%:*} ${w#*:} --check 16 > /dev/null || exit 7
done
python - <<PY
import json
for name in ("c2", "c2_k100"):
    d = json.load(open("gpurun_out/${tag}_%s_bench.json" % name))
    print(name, round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], d["dense_reasons"], "roofline %.3f" % d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "x ref floor %.2f" % d["speedup_over_reference_hbm_floor"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), "surface", (d.get("surface") or {}).get("pairs_per_s"))
PY
echo R05_FINAL_A_OK
