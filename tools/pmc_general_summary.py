#!/usr/bin/env python3
"""Means per launch of the PMC counters collected by tools/pmc_general.sh, per kernel; for k_classify9_group the launches are
split by strategy through their order in bench.py (hybrid legs first, then MRTL).  Writes profiles/<tag>.json."""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = {}
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    per = {}
    for r in rows:
        n = r["Kernel_Name"]
        k = "k_classify9_group" if "k_classify9_group" in n else "k_classify9_general" if "k_classify9_general" in n else "k_classify9" if "k_classify9<" in n else None
        if k:
            per.setdefault((k, r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), []).append(float(r["Counter_Value"]))
    for (k, c), disp in per.items():
        vals = [sum(disp[d]) for d in sorted(disp)]
        if k == "k_classify9_group" and len(vals) >= 6:     # bench.py: 1 + n_o launches per strategy, hybrid then mrtl
            h = len(vals) // 2
            acc.setdefault(k + " (hybrid legs)", {})[c] = sum(vals[:h]) / h
            acc.setdefault(k + " (mrtl legs)", {})[c] = sum(vals[h:]) / (len(vals) - h)
        else:
            acc.setdefault(k, {})[c] = sum(vals) / len(vals)
for k, d in acc.items():
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        d["_wait_fraction"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)
json.dump({"_doc": "tools/pmc_general.sh: rocprofv3 --pmc (one set per pass, --kernel-trace only) over bench.py --steps 2 --batches 2; mean per launch (10 M pairs).",
           "counters": acc}, open(os.path.join(root, "profiles", tag + ".json"), "w"), indent=1)
for k, d in acc.items():
    print(k, {c: ("%.3g" % v) for c, v in sorted(d.items())})
