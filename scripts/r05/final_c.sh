#!/bin/bash
# Round 5 reference numbers on the LAST sources of the round (the wide geometry's epoch length changed after final_a / final_b),
# one box, in the order of what must not be lost: the whole GPU suite, kernel-trace stats + PMC passes + bench line of C2 (the
# narrow geometry), then -- while the box's time lasts -- PMC passes + bench line of the C5 shard (the wide geometry, the one
# that changed).  Usage (gpurun): bash scripts/r05/final_c.sh [tag] [seconds after which no further step is started]
tag=${1:-r05_head}
limit=${2:-780}
start=$SECONDS
mkdir -p gpurun_out
timeout -k 10 560 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/${tag}_suite.log 2>&1
status=$?
tail -12 gpurun_out/${tag}_suite.log
[ $status -ne 0 ] && exit $status
echo "suite done at $((SECONDS - start)) s"
DS_BENCH_SURFACE=0 bash scripts/profile_bench.sh ${tag} > gpurun_out/${tag}_kernel_stats.txt 2>&1 || { tail -5 gpurun_out/${tag}_kernel_stats.txt; exit 2; }
export DS_BENCH_SURFACE=0
bash scripts/profile_pmc.sh ${tag}_c2 > gpurun_out/${tag}_pmc_c2.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c2.txt; exit 3; }
unset DS_BENCH_SURFACE
echo "C2 counters done at $((SECONDS - start)) s"
timeout -k 10 300 python bench.py > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.log || { tail -3 gpurun_out/${tag}_c2_bench.log; exit 5; }
echo "C2 bench done at $((SECONDS - start)) s"
if [ $((SECONDS - start)) -lt $limit ]; then
  DS_BENCH_SURFACE=0 timeout -k 10 400 python bench.py --config C5 --queries 125000 --steps 2 --warmup 1 --cpu-seconds 0 --check 8 > gpurun_out/${tag}_c5shard_bench.json 2> gpurun_out/${tag}_c5shard_bench.log || { tail -3 gpurun_out/${tag}_c5shard_bench.log; exit 8; }
  echo "C5 shard bench done at $((SECONDS - start)) s"
fi
if [ $((SECONDS - start)) -lt $limit ]; then
  DS_BENCH_SURFACE=0 bash scripts/profile_pmc.sh ${tag}_c5 --config C5 --queries 125000 > gpurun_out/${tag}_pmc_c5.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc_c5.txt; exit 4; }
  echo "C5 counters done at $((SECONDS - start)) s"
fi
python - <<PY
import json, os
for name in ("c2", "c5shard"):
    f = "gpurun_out/${tag}_%s_bench.json" % name
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    r = d["roofline"]
    print(name, d["build_id"], round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], "frac %.3f" % r["frac"], "traffic", r["traffic"],
          r.get("limited_by"), (d.get("cpu_baseline") or {}).get("value"), "surface", (d.get("surface") or {}).get("pairs_per_s"))
PY
echo R05_FINAL_C_OK
