#!/usr/bin/env python3
"""profiles/<tag>_stats.csv and profiles/<tag>_pmc.json from tools/pmc_prot.sh: counters of k_classify_prot9 per launch, split
into the plain and the seedextend legs of tools/bench_surface_b.py by launch order (1 + 5 launches each)."""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.exists(os.path.join(src, "kernel_stats.csv")):
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(root, "profiles", tag + "_stats.csv"))
bench = None
for line in open(os.path.join(src, "stats.log"), errors="replace"):
    if line.startswith("{") and "classify_prot9" in line:
        bench = json.loads(line)
acc = {}
for path in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(path)):
        if "k_classify_prot9" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), []).append(float(r["Counter_Value"]))
    for c, disp in per.items():
        vals = [sum(disp[d]) for d in sorted(disp)]
        h = len(vals) // 2
        acc[c] = {"plain": sum(vals[:h]) / max(1, h), "with_seedextend": sum(vals[h:]) / max(1, len(vals) - h), "launches": len(vals)}
out = {"_doc": "tools/pmc_prot.sh: rocprofv3 --pmc (one counter set per pass, --kernel-trace only) over tools/bench_surface_b.py; mean per launch of "
               "k_classify_prot9 (10 M pairs = 840 M windows).", "bench_line_under_stats_pass": bench, "counters": acc}
if "TCC_EA0_RDREQ_sum" in acc:
    out["L2_miss_requests_per_window"] = round(acc["TCC_EA0_RDREQ_sum"]["plain"] / 840e6, 3)
if "SQ_WAIT_ANY" in acc and "SQ_WAVE_CYCLES" in acc:
    out["wait_fraction"] = round(acc["SQ_WAIT_ANY"]["plain"] / acc["SQ_WAVE_CYCLES"]["plain"], 3)
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("counters", "_doc", "bench_line_under_stats_pass")}))
