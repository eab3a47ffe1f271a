#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_passes.sh into the summaries kept under profiles/:
   <tag>_kernel_stats.csv (the --stats table), <tag>_pmc_classify.json (mean counter values per launch of the
   fused kernel) and traffic.json (HBM bytes per launch: FETCH_SIZE + WRITE_SIZE, KiB -> B, separate passes;
   MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 64-byte requests, exact for the bucket gathers)."""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(prof, tag + "_kernel_stats.csv"))
counters = {}
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            for k in ("k_classify9", "k_classify9_group", "k_classify9_general", "k_classify_prot9", "k_lookup9", "k_agg_lca"):
                if k + "(" in name or k + "<" in name:
                    counters.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in counters.items()}
launches = {k: {c: len(v) for c, v in cs.items()} for k, cs in counters.items()}
doc = ("rocprofv3 --pmc passes (one counter set per run, with --kernel-trace only), mean per launch over the launches of "
       "bench.py --steps 2 --warmup 1 --batches 2 (headline and seedextend variants; 10 M pairs per launch). FETCH_SIZE/WRITE_SIZE in KiB.")
json.dump({"_doc": doc, "counters": summary, "launches": launches}, open(os.path.join(prof, tag + "_pmc_classify.json"), "w"), indent=1)
kc = summary.get("k_classify9", {})
if "FETCH_SIZE" in kc and "WRITE_SIZE" in kc:
    fetch, write = kc["FETCH_SIZE"] * 1024.0, kc["WRITE_SIZE"] * 1024.0
    bench_line = None
    log = os.path.join(src, "stats.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                bench_line = json.loads(line)
    tj = {"_doc": "HBM bytes per k_classify9 launch from rocprofv3 PMC (FETCH_SIZE + WRITE_SIZE, KiB->B; separate passes). FETCH_SIZE counts "
                  "64-B requests: exact for the 64-B bucket gathers; the coalesced 6.7 GB key stream may be tallied at half (MI355X_MICROARCH.md, HBM section).",
          "kernel": "k_classify9", "tag": tag, "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write)}
    if bench_line:
        tj["pairs_per_step"] = bench_line["config"].get("pairs_per_step_per_gpu")
        tj["n_keys"] = bench_line["config"].get("n_keys")
        tj["load_factor"] = bench_line["config"].get("load_factor")
    json.dump(tj, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
print(json.dumps({"kernels": list(summary), "stats_csv": bool(stats)}))
