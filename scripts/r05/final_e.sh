#!/bin/bash
# Last check of the round's last sources, the way the driver runs them: smoke(), the default bench line, and the driver's
# launcher line for N = 2 as a rehearsal on this one-GPU box (both ranks on the one device, DS_BENCH_SAME_DEVICE=1; RCCL cannot
# connect two ranks of one device, so the gather goes through the host and the line says so).
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_head_smoke.log 2>&1 || { tail -5 gpurun_out/r05_head_smoke.log; exit 1; }
tail -2 gpurun_out/r05_head_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r05_head_default_bench.json 2> gpurun_out/r05_head_default_bench.log || { tail -3 gpurun_out/r05_head_default_bench.log; exit 2; }
DS_BENCH_SAME_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --allow-host-fallback \
    > gpurun_out/r05_head_dist_rehearsal_bench.json 2> gpurun_out/r05_head_dist_rehearsal_bench.log || { tail -5 gpurun_out/r05_head_dist_rehearsal_bench.log; exit 3; }
python - <<PY
import json
for name in ("default", "dist_rehearsal"):
    d = json.load(open("gpurun_out/r05_head_%s_bench.json" % name))
    print(name, d["build_id"], "n_gpus", d["n_gpus"], round(d["value"]), "ms/step %.2f" % d["ms_per_step"], "verified", d.get("verified_queries"), "rccl_ranks", d.get("rccl_ranks"), d.get("scaling"))
PY
echo R05_FINAL_E_OK
