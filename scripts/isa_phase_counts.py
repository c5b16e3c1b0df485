#!/usr/bin/env python3
"""Static instruction counts of ds_jaccard_topk_kernel<false, false> by source phase (VERDICT r02, next-round item 3 ii).

Compiles one geometry to ISA with line tables (hipcc -S -gline-tables-only --cuda-device-only), attributes every
instruction of the kernel to the last line of ds_jaccard_impl.inc named by a `.loc`, and sums by the phases below (source
line ranges found from marker comments, so the table follows the file).  STATIC counts: a loop body counts once.

usage: isa_phase_counts.py narrow|wide [output.txt]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "doppel-speller_amd", "csrc")

# (phase name, marker that STARTS it) in file order; a phase ends where the next begins
MARKERS = [
    ("setup of a query (tables, start tile)", "ds_jaccard_topk_kernel(const void *arguments)"),
    ("candidate append (flush_raw / append_raw)", "auto flush_raw = [&]() {"),
    ("tile loop head: epochs, list pointers, band test", "bool live[kRound];"),
    ("item map (build_map)", "// ---- work items of this tile."),
    ("locate + request (request_round)", "// one round of this wave's items of a tile"),
    ("scatter: rounds, atomics", "// ---- (1) scatter"),
    ("collect sweep + refinement", "// ---- (2s) collect sweep"),
    ("dense scan + bootstrap", "// ---- (2d) dense scan"),
    ("tighten / select", "// ---- tighten"),
    ("exact stage + output", "// ---- exact evaluation."),
    ("(after the kernel)", "// ---- a column listed twice in a query"),
]

KINDS = [
    ("s_waitcnt", re.compile(r"^s_waitcnt")),
    ("s_nop", re.compile(r"^s_nop")),
    ("s_barrier", re.compile(r"^s_barrier")),
    ("branch", re.compile(r"^s_(c?branch|setpc|swappc)")),
    ("s_load (SMEM)", re.compile(r"^s_(load|buffer_load)")),
    ("SALU", re.compile(r"^s_")),
    ("lane spill (v_readlane/writelane)", re.compile(r"^v_(readlane|writelane)")),
    ("readfirstlane", re.compile(r"^v_readfirstlane")),
    ("scratch", re.compile(r"^scratch_")),
    ("LDS", re.compile(r"^ds_")),
    ("VMEM", re.compile(r"^(global|buffer|flat)_")),
    ("VALU", re.compile(r"^v_")),
]


def main():
    geometry = sys.argv[1]
    source_lines = open(os.path.join(CSRC, "ds_jaccard_impl.inc")).read().splitlines()
    starts = []
    for name, marker in MARKERS:
        line = next(i + 1 for i, text in enumerate(source_lines) if marker in text)
        starts.append((line, name))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        '-DDS_BUILD_ID="isa"', "-I", os.path.join(ROOT, "include"), "-gline-tables-only", "-S",
                        "--cuda-device-only", os.path.join(CSRC, f"ds_jaccard_{geometry}.hip"), "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    impl_file = None
    for line in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"ds_jaccard_impl\.inc"', line)
        if m:
            impl_file = m.group(1)
            break
    begin = next(i for i, line in enumerate(text) if re.match(r"^_ZN2ds\d+\w+22ds_jaccard_topk_kernelILb0ELb0EEEvPKv:", line))
    end = next(i for i in range(begin, len(text)) if text[i].strip().startswith(".end_amdhsa_kernel") or
               text[i].startswith(".Lfunc_end"))
    counts = collections.defaultdict(collections.Counter)
    current = starts[0][0]
    for line in text[begin:end]:
        stripped = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", stripped)
        if m:
            # helpers defined outside the kernel body (inlined) stay with the phase that called them
            if m.group(1) == impl_file and starts[0][0] <= int(m.group(2)) < starts[-1][0]:
                current = int(m.group(2))
            continue
        if not stripped or stripped[0] in ".;" or stripped.endswith(":"):
            continue
        phase = starts[0][1]
        for start, name in starts:
            if current >= start:
                phase = name
        op = stripped.split()[0]
        kind = next((k for k, pattern in KINDS if pattern.match(op)), "other")
        counts[phase][kind] += 1
    kinds = [k for k, _ in KINDS] + ["other"]
    header = f"{'phase':52s}" + "".join(f"{k.split(' ')[0][:9]:>10s}" for k in kinds) + f"{'all':>8s}"
    lines = [f"static instruction counts, {geometry} geometry, ds_jaccard_topk_kernel<false, false> "
             f"(scripts/isa_phase_counts.py; a loop body counts once)", header]
    total = collections.Counter()
    for _, name in starts[:-1]:
        row = counts[name]
        total.update(row)
        lines.append(f"{name:52s}" + "".join(f"{row[k]:10d}" for k in kinds) + f"{sum(row.values()):8d}")
    lines.append(f"{'whole kernel':52s}" + "".join(f"{total[k]:10d}" for k in kinds) + f"{sum(total.values()):8d}")
    meta = [line.strip() for line in text[end:end + 60] if re.search(r"; (codeLenInByte|NumSgprs|NumVgprs|ScratchSize|"
            r"Occupancy|LDSByteSize)", line)]
    report = "\n".join(lines + meta[:8])
    print(report)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as handle:
            handle.write(report + "\n")


if __name__ == "__main__":
    main()
