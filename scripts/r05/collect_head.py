"""Copies the measurements of scripts/r05/final_c.sh + final_d.sh (taken on the round's LAST sources, tag r05_head in gpurun_out/)
into profiles/ under the names of the round's reference set (profiles/r05_final_*), replacing the set of the build before the
wide geometry's epoch length changed.  The PMC entries of this build replace those of the same workload in
profiles/pmc_latest.json.  Usage: python scripts/r05/collect_head.py [tag in gpurun_out] [name in profiles]"""
import glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from doppel_speller_amd import _lib
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_head"
out = sys.argv[2] if len(sys.argv) > 2 else "r05_final"
bid = _lib.source_id()
path = os.path.join(root, "profiles/pmc_latest.json")
entries = {(e.get("queries"), e.get("truth"), e.get("k")): e for e in json.load(open(path))["entries"]}
for f in sorted(glob.glob(os.path.join(root, f"gpurun_out/pmc_{tag}_*_latest.json"))):
    for e in json.load(open(f)).get("entries", []):
        if e.get("build_id") == bid:
            entries[(e.get("queries"), e.get("truth"), e.get("k"))] = e
# the entries name the summary they were derived from by the tag of the run: point them at the copies under profiles/
text = json.dumps({"entries": list(entries.values())}, indent=1)
text = text.replace(f"profiles/{tag}_k100_pmc", f"profiles/{out}_c2_k100_pmc").replace(f"profiles/{tag}_", f"profiles/{out}_")
open(path, "w").write(text)
print("pmc_latest:", [(k, e["build_id"]) for k, e in entries.items()])
for w, name in (("c2", "c2"), ("k100", "c2_k100"), ("c3", "c3"), ("c5", "c5")):
    source = os.path.join(root, f"gpurun_out/pmc_{tag}_{w}_summary.txt")
    if os.path.exists(source):
        summary = open(source).read().replace(f"profiles/{tag}_k100_pmc", f"profiles/{out}_c2_k100_pmc").replace(f"profiles/{tag}_", f"profiles/{out}_")
        open(os.path.join(root, f"profiles/{out}_{name}_pmc_summary.txt"), "w").write(summary)
for w in ("c2", "k100", "c3s", "c5s"):
    source = os.path.join(root, f"gpurun_out/phase_{tag}_{w}_table.txt")
    if os.path.exists(source) and os.path.getsize(source):
        shutil.copy(source, os.path.join(root, f"profiles/r05_phase_table_{w}.txt"))
for name in ("c2", "c2_k100", "c3", "c5shard"):
    source = os.path.join(root, f"gpurun_out/{tag}_{name}_bench.json")
    if os.path.exists(source) and os.path.getsize(source):
        line = open(source).read().replace(f"profiles/{tag}_k100_pmc", f"profiles/{out}_c2_k100_pmc").replace(f"profiles/{tag}_", f"profiles/{out}_")
        open(os.path.join(root, f"profiles/{out}_{name}_bench.json"), "w").write(line)
        d = json.load(open(source))
        r = d["roofline"]
        assert d["build_id"] == bid, (d["build_id"], bid)
        print(name, round(d["value"]), "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in d["stages_ms"].items()},
              "frac %.3f ctr %s" % (r["frac"], r.get("frac_on_counter_traffic")), r.get("limited_by"), "| cpu",
              (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), "surface", (d.get("surface") or {}).get("pairs_per_s"))
stats = glob.glob(os.path.join(root, f"gpurun_out/prof_{tag}/**/*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(root, f"profiles/{out}_kernel_stats.csv"))
suite = os.path.join(root, f"gpurun_out/{tag}_suite.log")
if os.path.exists(suite):
    lines = open(suite).read().splitlines()
    open(os.path.join(root, f"profiles/{out}_gpu_suite.txt"), "w").write("\n".join([f"build {bid}: python -m pytest tests -x -q -m gpu --durations=8"] + lines[-14:]) + "\n")
print("build", bid)
