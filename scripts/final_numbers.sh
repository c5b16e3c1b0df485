#!/bin/bash
# The round's reference numbers on one GPU box: default bench (C2, with the CPU baseline), top-100, C3 at full size.
tag=${1:-r02_final}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.log || { tail -3 gpurun_out/${tag}_c2_bench.log; exit 1; }
timeout -k 10 300 python bench.py --k 100 --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/${tag}_c2_k100_bench.json 2> gpurun_out/${tag}_c2_k100_bench.log || exit 2
timeout -k 10 600 python bench.py --config C3 --steps 2 --warmup 1 --cpu-seconds 15 --check 32 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.log || { tail -3 gpurun_out/${tag}_c3_bench.log; exit 3; }
python - <<PY
import json
for name in ("c2", "c2_k100", "c3"):
    d = json.load(open("gpurun_out/${tag}_%s_bench.json" % name))
    print(name, round(d["value"]), "ms/step %.2f" % d["ms_per_step"], d["stages_ms"], d["dense_reasons"], "roofline %.3f" % d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "x ref floor %.2f" % d["speedup_over_reference_hbm_floor"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
