#!/usr/bin/env python3
"""Which stage of the unchanged four-process chain (scripts/umgap-analyse.sh:299-304) sets its pace?  Runs the stages one after
another over files (each alone on the machine) and prints seconds, MB/s of input and pairs/s per stage.  Inputs as in
tools/end_to_end.py.    python tools/stage_times.py [n_pairs]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import end_to_end as E

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
work = "/tmp/umgap_stages"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, n_index_keys = E._make_inputs(work, n_pairs, 250_000)
stages = [("prot2kmer2lca -o", ["prot2kmer2lca", "-o", idx]), ("seedextend -g1 -s3", ["seedextend", "-g1", "-s3"]), ("uniq -d /", ["uniq", "-d", "/"]),
          ("taxa2agg -l2 -a lca*", ["taxa2agg", "-l2", "-a", "lca*", tsv])]
res = {"pairs": n_pairs, "stages": []}
src = fasta
for i, (name, args) in enumerate(stages):
    dst = os.path.join(work, "stage%d.out" % i)
    # start-up (index load, taxonomy build) measured on an empty input and subtracted
    t0 = time.perf_counter()
    subprocess.run([E.UMGAP] + args, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, check=True)
    t_start = time.perf_counter() - t0
    t0 = time.perf_counter()
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        subprocess.run([E.UMGAP] + args, stdin=fi, stdout=fo, check=True)
    dt = time.perf_counter() - t0
    run = max(dt - t_start, 1e-9)
    res["stages"].append({"stage": name, "s": round(dt, 3), "start_up_s": round(t_start, 3), "in_MB": round(os.path.getsize(src) / 1e6, 1),
                          "out_MB": round(os.path.getsize(dst) / 1e6, 1), "in_MB_per_s": round(os.path.getsize(src) / 1e6 / run, 1),
                          "pairs_per_s": round(n_pairs / run, 1)})
    src = dst
print(json.dumps(res))
