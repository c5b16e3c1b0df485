#!/bin/bash
# Tuning sweep.  build: compile one library per variant into variants/ (here, no GPU needed).  run: on the GPU box,
# one short bench per variant, results appended to gpurun_out/sweep.txt.
#   scripts/sweep_variants.sh build name1="-DFLAG=1 ..." name2="..."
#   scripts/sweep_variants.sh run [bench args]
set -e
mode=$1; shift
if [ "$mode" = build ]; then
    mkdir -p variants
    for spec in "$@"; do
        name=${spec%%=*}; flags=${spec#*=}
        DS_LIBRARY=$PWD/variants/lib_$name.so DS_BUILD_FLAGS="$flags" python -c "import doppel_speller_amd as d; d.build_library(force=True)" 2>&1 | grep -v warning | tail -2
        echo "built $name ($flags)"
    done
else
    mkdir -p gpurun_out
    for so in variants/lib_*.so; do
        name=$(basename $so .so)
        DS_LIBRARY=$PWD/$so timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --check 16 "$@" > gpurun_out/sweep_$name.json 2> gpurun_out/sweep_$name.log || { echo "$name FAILED"; tail -3 gpurun_out/sweep_$name.log; exit 1; }
        python - "$name" <<'PY' | tee -a gpurun_out/sweep.txt
import json, sys
d = json.load(open("gpurun_out/sweep_%s.json" % sys.argv[1]))
s = d["stages_ms"]
print("%-28s topk %.2f ms dense %.2f ms total %.2f  dense queries %d" % (sys.argv[1], s["ds_jaccard_topk_kernel"], s["ds_jaccard_dense_kernel"], s["jaccard_topk"], d["dense_path_queries"]))
PY
    done
fi
