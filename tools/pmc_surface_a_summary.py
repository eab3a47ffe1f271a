#!/usr/bin/env python3
"""gpurun_out/<tag>/ of tools/pmc_surface_a.sh -> profiles/<name>.json: read requests per k-mer of the LAST launch of each kernel
instantiation (legs in bench_surface_a_walk.py's order: plain no-fragments, plain fragments, walk no-fragments, walk fragments).
    python tools/pmc_surface_a_summary.py <tag> <name>"""
import csv
import json
import os
import sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.join("gpurun_out", tag)
res = {"what": "L2 -> fabric read requests (TCC_EA0_RDREQ_sum, one rocprofv3 --pmc pass) per k-mer of umgap_classify9_dev over the plain table "
               "(k_classify9<true, 0>) and over the same proteome's walk-order index (k_classify9<true, 2>): 10 M pairs of 84 packed 9-mers cut from a "
               "2.1 G-window proteome, bound tables; tools/pmc_surface_a.sh.  The figure includes the key stream itself (8 B per k-mer = 0.125 requests).",
       "runs": {}}
for m in (0, 30):
    rows = list(csv.DictReader(open(os.path.join(root, "m%d" % m, "pmc_counter_collection.csv"))))
    by = {}
    for r in rows:
        if "k_classify9" in r["Kernel_Name"] and r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
            by.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    seq = [by[k] for k in sorted(by)]
    bound_plain = [v for n, v in seq if "<true, 0>" in n]
    bound_walk = [v for n, v in seq if "<true, 2>" in n]
    kmers = 10_000_000 * 84
    # per layout the tool runs: leg 1 warm-up + timed, leg 2 warm-up + timed -> the last launch of each leg
    line = json.loads(open(os.path.join(root, "m%d.json" % m)).read().strip().splitlines()[-1])
    res["runs"]["%.1f %% of the residues replaced" % (m / 10.0)] = {
        "requests_per_kmer_plain_table": round(bound_plain[-1] / kmers, 3),
        "requests_per_kmer_walk_index_no_fragment_structure_given": round(bound_walk[len(bound_walk) // 2 - 1] / kmers, 3),
        "requests_per_kmer_walk_index_runs_cut_at_fragments": round(bound_walk[-1] / kmers, 3),
        "launches_seen": {"plain": len(bound_plain), "walk": len(bound_walk)},
        "timing_under_pmc": {k: line[k] for k in ("plain_table", "walk_order_index")}}
json.dump(res, open(os.path.join("profiles", name + ".json"), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "timing_under_pmc"} for k, v in res["runs"].items()}))
