"""Copies a round's reference measurements from gpurun_out/ into profiles/ (the PMC entries of the two boxes merged into
profiles/pmc_latest.json for the current build id) and prints the DESIGN.md tables.  Usage: python scripts/r05/collect_final.py r05_final"""
import glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from doppel_speller_amd import _lib
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_final"
bid = _lib.source_id()
entries = {}
for f in (f"gpurun_out/pmc_{tag}_k100_latest.json", f"gpurun_out/pmc_{tag}_c5_latest.json"):
    d = json.load(open(os.path.join(root, f)))
    for e in d.get("entries", [d]):
        if e.get("build_id") == bid:
            entries[(e.get("queries"), e.get("truth"), e.get("k"))] = e
assert len(entries) == 4, (bid, sorted(entries))
json.dump({"entries": list(entries.values())}, open(os.path.join(root, "profiles/pmc_latest.json"), "w"), indent=1)
for w, name in (("c2", "c2"), ("k100", "c2_k100"), ("c3", "c3"), ("c5", "c5")):
    shutil.copy(os.path.join(root, f"gpurun_out/pmc_{tag}_{w}_summary.txt"), os.path.join(root, f"profiles/{tag}_{name}_pmc_summary.txt"))
for name in ("c2", "c2_k100", "c3", "c5shard"):
    shutil.copy(os.path.join(root, f"gpurun_out/{tag}_{name}_bench.json"), os.path.join(root, "profiles"))
for w, name in (("c2", "c2"), ("k100", "k100"), ("c3s", "c3s"), ("c5s", "c5s")):
    shutil.copy(os.path.join(root, f"gpurun_out/phase_{tag}_{w}_table.txt"), os.path.join(root, f"profiles/r05_phase_table_{name}.txt"))
stats = max(glob.glob(os.path.join(root, f"gpurun_out/prof_{tag}/**/*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats, os.path.join(root, f"profiles/{tag}_kernel_stats.csv"))
print("build", bid)
for name in ("c2", "c2_k100", "c3", "c5shard"):
    d = json.load(open(os.path.join(root, f"profiles/{tag}_{name}_bench.json")))
    r = d["roofline"]
    assert d["build_id"] == bid
    print(name, round(d["value"]), "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in d["stages_ms"].items()}, "frac %.3f ctr %.3f" % (r["frac"], r["frac_on_counter_traffic"]),
          r["limited_by"], "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), "surface", (d.get("surface") or {}).get("pairs_per_s"))
