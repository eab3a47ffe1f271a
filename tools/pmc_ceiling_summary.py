#!/usr/bin/env python3
"""Summarises the passes of tools/pmc_ceiling.sh into profiles/<tag>.json: per kernel and counter the mean value per
launch; for counters collected per instance (no _sum) the per-instance spread (min / max / mean) = channel balance."""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_gather_mlp", "k_gather_policy", "k_classify9")
acc = {}
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            k = next((k for k in KERNELS if k in name), None)
            if k is None:
                continue
            if k == "k_gather_mlp":
                k += "<4,8>" if ("4, 8" in name or "Li4ELi8" in name) else "<other>"
            key = (k, row["Counter_Name"])
            acc.setdefault(key, {}).setdefault(row.get("Dispatch_Id", "0"), []).append(float(row["Counter_Value"]))
out = {}
for (k, c), disp in sorted(acc.items()):
    per_launch = [sum(v) for v in disp.values()]
    e = {"mean_per_launch": sum(per_launch) / len(per_launch), "launches": len(per_launch)}
    inst = [len(v) for v in disp.values()]
    if max(inst) > 1:   # per-instance rows: balance across TCC channels / XCDs
        flat = [x for v in disp.values() for x in v]
        e["instances"] = max(inst)
        e["instance_min"], e["instance_max"], e["instance_mean"] = min(flat), max(flat), sum(flat) / len(flat)
    out.setdefault(k, {})[c] = e
logs = {}
for lp in sorted(glob.glob(os.path.join(src, "p*.log"))):
    for line in open(lp, errors="replace"):
        if line.startswith("{") and "table_GB" in line:
            logs[os.path.basename(lp)] = json.loads(line)
doc = {"_doc": "tools/pmc_ceiling.sh: rocprofv3 --pmc, one counter set per pass (with --kernel-trace only), over tools/ceiling_probe.py; "
               "mean per launch (840 M random 64-byte quad probes of the 128 GB table for the gather kernels).",
       "counters": out, "rates_under_profiler": logs}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(doc, open(os.path.join(root, "profiles", tag + ".json"), "w"), indent=1)
print(json.dumps({k: sorted(v) for k, v in out.items()}))
