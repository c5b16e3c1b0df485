#!/bin/bash
# Round 5 reference numbers on the round's LAST sources, second box (after final_c.sh): PMC passes + bench lines of C2 at the
# reference's prediction top_n = 100 and of C3, then the phase tables of the diagnostics build (variants/lib_diag.so, built from
# the same sources) on the four shapes.  Usage (gpurun): bash scripts/r05/final_d.sh [tag] [seconds after which no further step is started]
tag=${1:-r05_head}
limit=${2:-600}
start=$SECONDS
mkdir -p gpurun_out
export DS_BENCH_SURFACE=0
bash scripts/profile_pmc.sh ${tag}_k100 --k 100 > gpurun_out/${tag}_pmc_k100.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_k100.txt; exit 4; }
timeout -k 10 300 python bench.py --k 100 --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/${tag}_c2_k100_bench.json 2> gpurun_out/${tag}_c2_k100_bench.log || exit 6
echo "top-100 done at $((SECONDS - start)) s"
bash scripts/profile_pmc.sh ${tag}_c3 --config C3 > gpurun_out/${tag}_pmc_c3.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c3.txt; exit 3; }
unset DS_BENCH_SURFACE
timeout -k 10 300 python bench.py --config C3 --steps 2 --warmup 1 --cpu-seconds 10 --check 32 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.log || { tail -3 gpurun_out/${tag}_c3_bench.log; exit 7; }
echo "C3 done at $((SECONDS - start)) s"
if [ -f variants/lib_diag.so ]; then
  for w in "c2:" "k100:--k 100" "c3s:--truth 5000000 --queries 50000 --k 50" "c5s:--config C5 --queries 20000"; do
    [ $((SECONDS - start)) -lt $limit ] || break
    DS_ALLOW_STALE_LIBRARY=1 DS_BENCH_SURFACE=0 bash scripts/phase_run.sh ${tag}_${w%%:*} ${w#*:} --check 16 > /dev/null || exit 8
    echo "phase table ${w%%:*} done at $((SECONDS - start)) s"
  done
fi
python - <<PY
import json
for name in ("c2_k100", "c3"):
    d = json.load(open("gpurun_out/${tag}_%s_bench.json" % name))
    r = d["roofline"]
    print(name, d["build_id"], round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], d["dense_reasons"], "frac %.3f" % r["frac"], "traffic", r["traffic"],
          r.get("limited_by"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
echo R05_FINAL_D_OK
