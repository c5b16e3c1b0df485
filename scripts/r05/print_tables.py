"""Prints the rows of DESIGN.md's state table and of §3's "what bounds the kernel" table from profiles/r05_final_*_bench.json and
profiles/pmc_latest.json.  Usage: python scripts/r05/print_tables.py"""
import json, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
entries = {(e["queries"], e["truth"], e["k"]): e for e in json.load(open(os.path.join(root, "profiles/pmc_latest.json")))["entries"]}
rows = []
for name in ("c2", "c2_k100", "c3", "c5shard"):
    d = json.load(open(os.path.join(root, f"profiles/r05_final_{name}_bench.json")))
    r, s = d["roofline"], d["stages_ms"]
    print(name, d["build_id"], "%.1fM pairs/s" % (d["value"] / 1e6), "%.2f ms" % d["ms_per_step"], "kernels %.2f + %.2f" % (s["ds_jaccard_topk_kernel"], s["ds_jaccard_dense_kernel"]),
          "features %.2f" % s["construct_features"], "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"),
          "surface", (d.get("surface") or {}).get("pairs_per_s"), "8d frac %.2f" % r.get("frac_on_survey_8d_bytes", 0), "redos", d.get("sparse_redos"))
    key = {"c2": (100000, 500000, 10), "c2_k100": (100000, 500000, 100), "c3": (1000000, 5000000, 50), "c5shard": (125000, 50000000, 100)}[name]
    e = entries.get(key)
    assert e is None or e["build_id"] == d["build_id"], (name, e["build_id"], d["build_id"])
    rows.append((name, d, e))
for name, d, e in rows:
    if e is None:
        print(name, "no counter entry"); continue
    b, r = e["bound_model"], d["roofline"]
    print(name, "| pmc %.2f ms | VALU/SALU %.1f / %.1f | issue %.2f | lds %.2f (%.2f) | hbm %.2f | wait %.2f | L2 %.2f | req/traffic %.1f / %.1f GB | frac %.3f | ctr %.3f | %s | %s" % (
        b["kernel_ms_under_pmc"], b["valu_issue_ms"], b["salu_issue_ms"], b["frac_issue"], b["frac_lds"], b["lds_bank_conflict_share"], b["frac_hbm"], b["wait_share_of_wave_cycles"],
        b["l2_hit_rate"], e["requested_bytes_per_launch"] / 1e9, e["hbm_bytes_per_launch"] / 1e9, r["frac"], r.get("frac_on_counter_traffic") or e["hbm_bytes_per_launch"] / (d["stages_ms"]["ds_jaccard_topk_kernel"] * 1e-3) / 8e12, r.get("memory_level"), r.get("limited_by")))
