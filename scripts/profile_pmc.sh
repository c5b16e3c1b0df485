#!/bin/bash
# PMC counter passes for the default bench workload (separate rocprofv3 runs, counters only + kernel trace).
# Usage (on the GPU box via gpurun): bash scripts/profile_pmc.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {  # name, counters...
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --check 0 > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.log
    echo "$name exit=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py ${tag} 100000 500000 10 --write | tee gpurun_out/pmc_${tag}_summary.txt
cp profiles/pmc_latest.json gpurun_out/pmc_${tag}_latest.json
