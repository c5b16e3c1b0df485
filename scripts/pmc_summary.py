#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes of scripts/profile_pmc.sh into (a) a text summary and (b) profiles/pmc_latest.json, the
file bench.py reads `roofline.traffic` and `roofline.bound_model` from (only when its build_id and workload match).

Bound model of ds_jaccard_topk_kernel.  Issue costs measured by scripts/micro/calibrate.hip on
MI355X (profiles/r02_calibration.txt), 4 waves per SIMD as in the kernel:
    VALU   2.53 cycles per wave-instruction per SIMD (SIMD-32: 2 cycles + issue gaps)
    SALU   1.09 cycles per instruction per CU (ONE scalar unit per CU: 4.38 cycles per SIMD when all four SIMDs issue)
    a (VALU, SALU) pair in one wave's stream issues in 6.11 cycles per SIMD ~ 2.53 + 0.82 * 4.38: the two barely overlap
FETCH_SIZE (KiB) reports exactly half of the bytes of coalesced streams at 4, 8 and 16 bytes per lane (same file), so
HBM traffic = 2 * FETCH_SIZE + WRITE_SIZE for this kernel, whose reads are coalesced posting / sums streams.

usage: pmc_summary.py <tag> [--write]      (workload = the `config` of the bench line of the fetch pass)

profiles/pmc_latest.json holds ONE ENTRY PER WORKLOAD ({"entries": [...]}, each with the build id it was measured on);
--write replaces the entry of this workload and keeps the others.
"""
import collections
import csv
import glob
import json
import os
import sys

VALU_CYCLES, SALU_CYCLES_PER_SIMD, PAIR_OVERLAP = 2.53, 4.38, 0.82
SIMDS, CUS, HBM_PEAK = 1024, 256, 8.0e12


def main():
    tag = sys.argv[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "gpurun_out", f"pmc_{tag}_fetch.json")) as handle:
        bench_line = json.loads(handle.read().strip().splitlines()[-1])
    queries, truth, k = (bench_line["config"][name] for name in ("queries_per_gpu", "truth_titles", "k"))
    print(f"workload: {bench_line['config']['workload']}")
    totals = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    durations = collections.defaultdict(list)
    for name in ("fetch", "write", "sq1", "sq2", "tcc"):
        base = os.path.join(root, "gpurun_out", f"pmc_{tag}_{name}")
        for path in glob.glob(base + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                kernel = row["Kernel_Name"].split("(")[0]
                totals[kernel][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[(kernel, row["Counter_Name"])] += 1
        for path in glob.glob(base + "/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                durations[row["Kernel_Name"].split("(")[0]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    per = {kernel: {c: v / calls[(kernel, c)] for c, v in counters.items()} for kernel, counters in totals.items()}
    for kernel, counters in sorted(per.items()):
        mean_ms = sum(durations[kernel]) / max(1, len(durations[kernel])) / 1e6
        print(f"{kernel}: {len(durations[kernel])} dispatches under PMC, mean {mean_ms:.3f} ms")
        for counter, value in sorted(counters.items()):
            print(f"    {counter:24s} {value:.6g} per dispatch")
    kernel = next((name for name in per if "ds_jaccard_topk_kernel<false" in name), None)
    if kernel is None:
        return
    c = per[kernel]
    t = sum(durations[kernel]) / len(durations[kernel]) / 1e9          # seconds, under the profiler
    clock = c["SQ_BUSY_CYCLES"] / 32 / t                                 # SQ_BUSY_CYCLES sums the 32 shader engines
    valu = c["SQ_INSTS_VALU"] * VALU_CYCLES / SIMDS / clock
    salu = c["SQ_INSTS_SALU"] * SALU_CYCLES_PER_SIMD / SIMDS / clock
    issue = valu + PAIR_OVERLAP * salu
    lds = c["SQ_LDS_IDX_ACTIVE"] / CUS / clock
    hbm_bytes = 2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
    hbm = hbm_bytes / HBM_PEAK
    model = {
        "kernel_ms_under_pmc": t * 1e3, "clock_ghz": clock / 1e9,
        "valu_issue_ms": valu * 1e3, "salu_issue_ms": salu * 1e3, "issue_ms": issue * 1e3,
        "lds_busy_ms": lds * 1e3, "hbm_ms_at_8TBs": hbm * 1e3,
        "frac_issue": issue / t, "frac_lds": lds / t, "frac_hbm": hbm / t,
        "binding": max((("instruction issue (VALU + SALU)", issue), ("LDS", lds), ("HBM", hbm)), key=lambda x: x[1])[0],
        "frac": max(issue, lds, hbm) / t,
        "wait_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        "insts": {name: c[name] for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                             "SQ_INSTS_SMEM")},
        "source": f"profiles/{tag}_pmc_summary.txt; issue costs and the FETCH_SIZE factor from profiles/r02_calibration.txt",
    }
    print(json.dumps(model, indent=1))
    # ---- ds_construct_features_kernel: the same model (4 workgroups of 4 waves per CU = 4 waves per SIMD, like the calibration)
    features_model = None
    features_kernel = next((name for name in per if "ds_construct_features_kernel" in name), None)
    if features_kernel is not None and durations[features_kernel]:
        f = per[features_kernel]
        tf = sum(durations[features_kernel]) / len(durations[features_kernel]) / 1e9
        f_clock = f["SQ_BUSY_CYCLES"] / 32 / tf
        f_valu = f["SQ_INSTS_VALU"] * VALU_CYCLES / SIMDS / f_clock
        f_salu = f["SQ_INSTS_SALU"] * SALU_CYCLES_PER_SIMD / SIMDS / f_clock
        f_issue = f_valu + PAIR_OVERLAP * f_salu
        f_lds = f["SQ_LDS_IDX_ACTIVE"] / CUS / f_clock
        f_hbm = (2.0 * f["FETCH_SIZE"] * 1024 + f["WRITE_SIZE"] * 1024) / HBM_PEAK
        pairs = queries * k
        features_model = {
            "kernel_ms_under_pmc": tf * 1e3, "clock_ghz": f_clock / 1e9, "pairs": pairs,
            "valu_issue_ms": f_valu * 1e3, "salu_issue_ms": f_salu * 1e3, "issue_ms": f_issue * 1e3,
            "lds_busy_ms": f_lds * 1e3, "hbm_ms_at_8TBs": f_hbm * 1e3,
            "frac_issue": f_issue / tf, "frac_lds": f_lds / tf, "frac_hbm": f_hbm / tf,
            "binding": max((("instruction issue (VALU + SALU)", f_issue), ("LDS", f_lds), ("HBM", f_hbm)), key=lambda x: x[1])[0],
            "frac": max(f_issue, f_lds, f_hbm) / tf,
            "wait_share_of_wave_cycles": f["SQ_WAIT_ANY"] / f["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_share": f["SQ_LDS_BANK_CONFLICT"] / f["SQ_LDS_IDX_ACTIVE"],
            "wave_instructions_per_pair": {name[len("SQ_INSTS_"):]: f[name] / pairs
                                           for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")},
            "where_the_vector_instructions_go": "profiles/r04_features_isa_counts.txt x the trip counts of the bench line "
                                                "(roofline_features.recurrence_steps_per_pair): the bit-parallel recurrence "
                                                "(5-10 VALU per text character) and the staging around it",
            "source": f"profiles/{tag}_pmc_summary.txt; issue costs from profiles/r02_calibration.txt (4 waves per SIMD)",
        }
        print("ds_construct_features_kernel:", json.dumps(features_model, indent=1))
    if "--write" in sys.argv:
        sys.path.insert(0, root)
        from doppel_speller_amd import _lib
        out = {"queries": queries, "truth": truth, "k": k, "kernel": kernel.split("<")[0],
               "geometry": bench_line["roofline"].get("geometry"), "tag": tag,
               "requested_bytes_per_launch": bench_line["roofline"].get("requested_bytes_per_launch"),
               "build_id": _lib.source_id(), "FETCH_SIZE_KiB_per_launch": c["FETCH_SIZE"],
               "WRITE_SIZE_KiB_per_launch": c["WRITE_SIZE"], "hbm_bytes_per_launch": hbm_bytes,
               "note": "hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): FETCH_SIZE counts half the bytes of "
                       "coalesced 4/8/16-byte-per-lane streams on gfx950 (profiles/r02_calibration.txt)",
               "bound_model": model, "features_bound_model": features_model}
        path = os.path.join(root, "profiles", "pmc_latest.json")
        entries = []
        if os.path.exists(path):
            with open(path) as handle:
                previous = json.load(handle)
            entries = [e for e in previous.get("entries", [previous])
                       if (e.get("queries"), e.get("truth"), e.get("k")) != (queries, truth, k)]
        with open(path, "w") as handle:
            json.dump({"entries": entries + [out]}, handle, indent=1)


if __name__ == "__main__":
    main()
