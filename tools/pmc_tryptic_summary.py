#!/usr/bin/env python3
"""profiles/<tag>_pmc.json from the output of tools/pmc_tryptic.sh: mean counters per launch of the kernels of the fused tryptic call
(k_tryp_*, k_lookup_var_pipe, k_agg_csr_group), per 10 M pairs = 40 M pieces."""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = {}
for path in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        for name in ("k_tryp_mark", "k_tryp_count", "k_tryp_scan", "k_tryp_write", "k_tryp_reads", "k_lookup_var_pipe", "k_agg_csr_group"):
            if name in k:
                per.setdefault((name, r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[(name, r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, c), disp in per.items():
        v = [disp[d] for d in sorted(disp)]
        acc.setdefault(name, {})[c] = sum(v) / len(v)
out = {"_doc": "tools/pmc_tryptic.sh: rocprofv3 --pmc (one counter set per pass, --kernel-trace only) over tools/bench_tryptic_fused.py 400000000 10000000; "
               "mean per launch (10 M pairs = 40 M pieces per launch).", "counters": acc}
for k in ("k_lookup_var_pipe", "k_agg_csr_group"):
    if k in acc and "TCC_EA0_RDREQ_sum" in acc[k]:
        out.setdefault("L2_miss_requests_per_launch", {})[k] = round(acc[k]["TCC_EA0_RDREQ_sum"])
    if k in acc and "SQ_WAIT_ANY" in acc[k] and "SQ_WAVE_CYCLES" in acc[k]:
        out.setdefault("share_of_wave_cycles_waiting", {})[k] = round(acc[k]["SQ_WAIT_ANY"] / acc[k]["SQ_WAVE_CYCLES"], 3)
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("counters", "_doc")}))
