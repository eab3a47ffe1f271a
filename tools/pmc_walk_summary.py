#!/usr/bin/env python3
"""Counters of the fused surface-B kernels per launch and kernel variant (k_classify_prot9<BOUND, GENERAL, WALK>) from a
tools/pmc_prot.sh output directory:  python tools/pmc_walk_summary.py gpurun_out/<tag>  -> one JSON object on stdout."""
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
acc = {}
for path in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_classify_prot9<([^>]*)>", r["Kernel_Name"])
        if m:
            per.setdefault((m.group(1).replace(" ", ""), r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), []).append(float(r["Counter_Value"]))
    for (variant, c), disp in per.items():
        vals = [sum(disp[d]) for d in sorted(disp)]
        acc.setdefault(variant, {})[c] = round(sum(vals) / len(vals), 1)
dur = {}
for path in glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_classify_prot9<([^>]*)>", r["Kernel_Name"])
        if m:
            dur.setdefault(m.group(1).replace(" ", ""), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for v, c in acc.items():
    if v in dur:
        c["ms_mean"] = round(sum(dur[v]) / len(dur[v]), 3)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        c["wait_fraction"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if "SQ_ACTIVE_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        c["issue_fraction"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if "TCC_EA0_RDREQ_sum" in c:
        c["L2_miss_requests_per_window"] = round(c["TCC_EA0_RDREQ_sum"] / 840e6, 3)
    if "SQ_INSTS_VALU" in c:
        c["valu_per_window"] = round(c["SQ_INSTS_VALU"] * 64 / 840e6, 1)     # wave instructions x 64 lanes / windows
print(json.dumps({"variants <BOUND,GENERAL,WALK>": acc}, indent=1))
