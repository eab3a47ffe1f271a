#!/usr/bin/env python3
"""End-to-end throughput of the `umgap` command on the GPU box: text in -> text out, the stages of
the 9-mer high-precision pipeline (scripts/umgap-analyse.sh:299-304 minus FGSpp) as separate
processes connected by pipes, exactly as the script runs them.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from umgap_amd import synth

UMGAP = os.path.join(ROOT, "umgap_amd", "bin", "umgap")
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/umgap_cli"
os.makedirs(work, exist_ok=True)
n_tax, n_keys = 100_000, 50_000_000
w = synth.SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
idx, tsv = os.path.join(work, "ninemer.umgapix"), os.path.join(work, "taxons.tsv")
w.index.save(idx)
synth.write_taxonomy_tsv(tsv, w.ids, w.parents, w.valid, w.rank)
keys = w.reads(0, n_pairs).cpu().numpy().view(np.uint64)
# pept2lca surface: one record per read end (42 9-mers, one per line)
letters = np.zeros((len(keys), 10), np.uint8)
for j in range(9):
    letters[:, j] = ((keys >> np.uint64(5 * (8 - j))) & np.uint64(31)).astype(np.uint8) + 65
letters[:, 9] = 10
body = letters.reshape(2 * n_pairs, 42 * 10)
fasta = os.path.join(work, "in.fa")
with open(fasta, "wb") as f:
    for r in range(2 * n_pairs):
        f.write(b">read%d/%d\n" % (r // 2, r % 2 + 1))
        f.write(body[r].tobytes())
size = os.path.getsize(fasta)
del w
torch.cuda.empty_cache()


def timed(cmd, stdin_path, stdout_path):
    t0 = time.perf_counter()
    with open(stdin_path, "rb") as fi, open(stdout_path, "wb") as fo:
        subprocess.run(cmd, stdin=fi, stdout=fo, check=True)
    return time.perf_counter() - t0


res = {"pairs": n_pairs, "input_MB": round(size / 1e6, 1)}
h, se, un, out = (os.path.join(work, x) for x in ("hits.fa", "se.fa", "un.fa", "out.fa"))
res["pept2lca_s"] = round(timed([UMGAP, "pept2lca", "-o", idx], fasta, h), 3)
res["seedextend_s"] = round(timed([UMGAP, "seedextend", "-g", "1", "-s", "3"], h, se), 3)
res["uniq_s"] = round(timed([UMGAP, "uniq", "-d", "/"], se, un), 3)
res["taxa2agg_s"] = round(timed([UMGAP, "taxa2agg", "-l", "2", "-a", "lca*", tsv], un, out), 3)
t0 = time.perf_counter()
pipe = "%s pept2lca -o %s < %s | %s seedextend -g1 -s3 | %s uniq -d / | %s taxa2agg -l2 -a 'lca*' %s > %s.pipe" % (UMGAP, idx, fasta, UMGAP, UMGAP, UMGAP, tsv, out)
subprocess.run(pipe, shell=True, check=True)
res["pipeline_s"] = round(time.perf_counter() - t0, 3)
res["pipeline_pairs_per_s"] = round(n_pairs / res["pipeline_s"], 1)
res["same_output"] = open(out, "rb").read() == open(out + ".pipe", "rb").read()
res["records_out"] = open(out, "rb").read().count(b">")
print(json.dumps(res))
