#!/bin/bash
# ThreadSanitizer over the threaded HOST builds of libdoppel_amd.so (ds_problem_create, ds_index_create's image, encoders, duplicate
# ranks: std::thread fan-outs whose outputs must not depend on the thread count): same recipe as scripts/sanitize_host.sh with
# -fsanitize=thread, 8 build threads.  CPU only.  Usage: bash scripts/tsan_host.sh   (writes profiles/r05_tsan_host.txt)
set -o pipefail
out=/tmp/ds_tsan; mkdir -p $out
id=$(python -c "import sys; sys.path.insert(0, '.'); from doppel_speller_amd import _lib; print(_lib.source_id())")
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer \
      "-DDS_BUILD_ID=\"$id\"" -I include -pthread -o $out/libdoppel_amd_tsan.so doppel-speller_amd/csrc/*.hip || exit 1
runtime=$(find /opt/rocm/lib/llvm -name "libclang_rt.tsan*x86_64*.so" | head -1)
LD_PRELOAD=$runtime TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 DS_HOST_THREADS=8 DS_LIBRARY=$out/libdoppel_amd_tsan.so DS_AUTO_REBUILD=0 \
timeout 1500 python -m pytest tests/test_native_build_cpu.py -q -p no:cacheprovider 2>&1 | tail -3 | tee profiles/r05_tsan_host.txt
