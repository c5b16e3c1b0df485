#!/bin/bash
# PMC counter passes of one bench configuration (separate rocprofv3 runs: counters + kernel trace only, never a
# sys/hip/hsa trace).  The program after `--` is python3 itself, one GPU, one process.
# Usage (on the GPU box via gpurun): bash scripts/profile_pmc.sh <tag> [bench.py arguments, e.g. --config C3]
tag=${1:-r03}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {  # name, counters...
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --check 0 "${BENCH_ARGS[@]}" > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.log
    status=$?
    echo "$name exit=$status"
    return $status
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS || exit 1
run tcc TCC_HIT_sum TCC_MISS_sum || exit 1
python3 scripts/pmc_summary.py ${tag} --write | tee gpurun_out/pmc_${tag}_summary.txt
cp profiles/pmc_latest.json gpurun_out/pmc_${tag}_latest.json
