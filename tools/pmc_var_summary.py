#!/usr/bin/env python3
"""profiles/<tag>_stats.csv and profiles/<tag>_pmc.json from the output of tools/pmc_var.sh: counters of k_lookup_var per
launch class (insertion-ordered queries / shuffled sample), with requests and cache accesses per lookup."""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = None
for line in open(os.path.join(src, "stats.log"), errors="replace"):
    if line.startswith("{") and "lookup_var" in line:
        bench = json.loads(line)
if os.path.exists(os.path.join(src, "kernel_stats.csv")):
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(root, "profiles", tag + "_stats.csv"))
n_q = bench["n_queries"] if bench else None
n_s = min(n_q, 50_000_000) if n_q else None
acc = {}
for path in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "k_lookup_var_pipe" in r["Kernel_Name"] or "k_lookup_varp" in r["Kernel_Name"]]
    by_counter = {}
    for r in rows:
        by_counter.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), []).append(float(r["Counter_Value"]))
    for c, disp in by_counter.items():
        per = [sum(disp[d]) for d in sorted(disp)][-8:]     # the last launches of the probe kernel, in order: 4 x ordered, 4 x shuffled
        if len(per) >= 8:                                   # (before them: the build's self check)
            acc[c] = {"ordered": sum(per[1:4]) / 3, "shuffled": sum(per[5:8]) / 3}
out = {"_doc": "tools/pmc_var.sh: rocprofv3 --pmc (one counter set per pass, --kernel-trace only) over tools/bench_config5.py; mean per launch of "
               "k_lookup_var for the insertion-ordered queries (%s lookups per launch) and the shuffled sample (%s)." % (n_q, n_s),
       "bench_line": bench, "counters": acc}
if n_q and "TCC_EA0_RDREQ_sum" in acc:
    out["L2_miss_requests_per_lookup"] = {"ordered": round(acc["TCC_EA0_RDREQ_sum"]["ordered"] / n_q, 3), "shuffled": round(acc["TCC_EA0_RDREQ_sum"]["shuffled"] / n_s, 3)}
if n_q and "TCP_TOTAL_CACHE_ACCESSES_sum" in acc:
    out["vector_cache_accesses_per_lookup"] = {"ordered": round(acc["TCP_TOTAL_CACHE_ACCESSES_sum"]["ordered"] / n_q, 2), "shuffled": round(acc["TCP_TOTAL_CACHE_ACCESSES_sum"]["shuffled"] / n_s, 2)}
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("counters", "_doc")})[:600])
