#!/usr/bin/env python3
"""Static instruction counts of ds_construct_features_kernel by source phase.

Compiles csrc/ds_features.hip to ISA with line tables, attributes every instruction of the kernel to the last line of
ds_features.hip named by a `.loc` (helpers inlined from the same file count at their own lines), and sums by the phases
below.  STATIC counts: a loop body counts once; the dynamic picture needs the trip counts (printed by
scripts/features_bound_model.py from a workload sample).

usage: features_isa_counts.py [output.txt]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE = os.path.join(ROOT, "doppel-speller_amd", "csrc", "ds_features.hip")

# (phase, marker that starts it); helpers first (file order), then the kernel's sections
MARKERS = [
    ("(prologue)", "namespace ds {"),
    ("match masks (build_masks_g)", "__device__ __forceinline__ void build_masks("),
    ("bit-parallel recurrence (lcs_bitparallel)", "__device__ __forceinline__ int lcs_bitparallel("),
    ("literal DP (levenshtein_literal*)", "__device__ uint8_t levenshtein_literal("),
    ("small-alphabet test", "__device__ __forceinline__ bool codes_below_64("),
    ("literal DP (levenshtein_literal*)", "__device__ uint8_t levenshtein_literal_g("),
    ("small-alphabet test", "__device__ __forceinline__ bool codes_below_64_g("),
    ("levenshtein_g dispatch + ratio (float64)", "__device__ uint8_t levenshtein_g("),
    ("ratio table (per workgroup)", "__global__ __launch_bounds__(kFeatWaves * 64) void ds_construct_features_kernel"),
    ("pair loop head, indexes, lengths", "for (int64_t first_pair = wave_global * 2;"),
    ("stage strings, squeeze spaces", "// stage both strings; count spaces"),
    ("word boundaries", "// truth word boundaries"),
    ("lev(title, truth) call site", "const uint8_t lev_ratio ="),
    ("word loop: set-up, masks call, window loop control", "// ---- truth words loop"),
    ("word loop: half-wave max-reduce + best window", "int key = start < lw ?"),
    ("word loop: reconstructed title, features", "const int best_ratio = best_key >> 8;"),
    ("idf_s = float32(log(float64))", "// :153  idf_s of every word at once"),
    ("lev(reconstructed, truth) call site", "// :161-162  strip the first and the last space"),
    ("ranks, basic features, output", "// :158  ranks = 1 +"),
    ("(after the kernel)", "// fast_levenshtein_ratio for independent pairs"),
]
KINDS = [("s_waitcnt", r"^s_waitcnt"), ("s_nop", r"^s_nop"), ("branch", r"^s_(c?branch|setpc|swappc)"), ("SMEM", r"^s_(load|buffer_load)"),
         ("SALU", r"^s_"), ("LDS", r"^ds_"), ("VMEM", r"^(global|buffer|flat|scratch)_"), ("VALU", r"^v_")]


def main():
    source_lines = open(SOURCE).read().splitlines()
    starts = []
    for name, marker in MARKERS:
        line = next(i + 1 for i, text in enumerate(source_lines) if marker in text)
        starts.append((line, name))
    starts.sort()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "f.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", '-DDS_BUILD_ID="isa"',
                        "-I", os.path.join(ROOT, "include"), "-gline-tables-only", "-S", "--cuda-device-only", SOURCE, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    own = {m.group(1) for line in text for m in [re.match(r'\s*\.file\s+(\d+)\s+.*ds_features\.hip"', line)] if m}
    begin = next(i for i, line in enumerate(text) if re.match(r"^_ZN2ds28ds_construct_features_kernel", line))
    end = next(i for i in range(begin, len(text)) if text[i].startswith(".Lfunc_end"))
    counts = collections.defaultdict(collections.Counter)
    current = starts[0][0]
    for line in text[begin:end]:
        stripped = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", stripped)
        if m:
            if m.group(1) in own and int(m.group(2)) > 0:
                current = int(m.group(2))
            continue
        if not stripped or stripped[0] in ".;" or stripped.endswith(":"):
            continue
        phase = starts[0][1]
        for start, name in starts:
            if current >= start:
                phase = name
        op = stripped.split()[0]
        counts[phase][next((k for k, pattern in KINDS if re.match(pattern, op)), "other")] += 1
    kinds = [k for k, _ in KINDS] + ["other"]
    lines = ["static instruction counts of ds_construct_features_kernel (scripts/features_isa_counts.py; a loop body counts once)",
             f"{'phase':56s}" + "".join(f"{k:>10s}" for k in kinds) + f"{'all':>8s}"]
    total = collections.Counter()
    seen = []
    for _, name in starts:
        if name in seen or name.startswith("(after"):
            continue
        seen.append(name)
        row = counts[name]
        total.update(row)
        lines.append(f"{name:56s}" + "".join(f"{row[k]:10d}" for k in kinds) + f"{sum(row.values()):8d}")
    lines.append(f"{'whole kernel':56s}" + "".join(f"{total[k]:10d}" for k in kinds) + f"{sum(total.values()):8d}")
    lines += [line.strip() for line in text[end:end + 60]
              if re.search(r"; (codeLenInByte|TotalNumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize)", line)][:8]
    report = "\n".join(lines)
    print(report)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as handle:
            handle.write(report + "\n")


if __name__ == "__main__":
    main()
