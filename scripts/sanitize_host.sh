#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of libdoppel_amd.so (index build, native problem build, encoders,
# transform_title, forest loader, argument checks): the library's sources compiled with -fsanitize=address,undefined for the host
# only (-fno-gpu-sanitize: GPU sanitizers are not available on this pool), the CPU tests that call into the library run against
# it.  CPU only, no GPU needed.  Usage: bash scripts/sanitize_host.sh   (writes profiles/r05_sanitize_host.txt)
set -o pipefail
out=/tmp/ds_asan; mkdir -p $out
id=$(python -c "import sys; sys.path.insert(0, '.'); from doppel_speller_amd import _lib; print(_lib.source_id())")
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize \
      -fno-omit-frame-pointer "-DDS_BUILD_ID=\"$id\"" -I include -pthread -o $out/libdoppel_amd_asan.so doppel-speller_amd/csrc/*.hip || exit 1
runtime=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan*x86_64*.so" | head -1)
LD_PRELOAD=$runtime ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
DS_LIBRARY=$out/libdoppel_amd_asan.so DS_AUTO_REBUILD=0 timeout 1500 python -m pytest tests/test_native_build_cpu.py tests/test_host_cpu.py \
    tests/test_transform_title_cpu.py tests/test_transform_property_cpu.py tests/test_forest_cpu.py tests/test_reference_vectors.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4 | tee profiles/r05_sanitize_host.txt
