#!/usr/bin/env python3
"""A/B of `umgap prot2kmer2agg` with the residues handed to the device call in the tokenizer threads' own page-locked buffers
(umgap_classify_records_sg; default) against gathered into one array first (UMGAP_FUSE_PIECES=0), alternating, same box, same
input; the command's own phase times (UMGAP_TIMING=1) and the output compared byte for byte.
    python tools/fused_pieces_ab.py [n_pairs] [rounds]"""
import hashlib
import json
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import end_to_end as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
work = "/tmp/umgap_e2e"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs(work, n, 250_000)
res = {"pairs": n, "runs": {"pieces": [], "gathered": [], "pieces_buffers_allocated_by_the_stream": []}}
digests = set()
for r in range(rounds):
    for name, val in (("gathered", "0"), ("pieces", "1"), ("pieces_buffers_allocated_by_the_stream", "1")):
        env = dict(os.environ, UMGAP_TIMING="1", UMGAP_FUSE_PIECES=val)
        if name.startswith("pieces_buffers"):
            env["UMGAP_FUSE_PREALLOC"] = "0"
        out = os.path.join(work, "out.ab")
        t0 = time.perf_counter()
        with open(fasta, "rb") as fi, open(out, "wb") as fo:
            p = subprocess.run([E.UMGAP, "prot2kmer2agg", "-g", "1", "-s", "3", "-d", "/", "-l", "2", "-a", "lca*", idx, tsv], stdin=fi, stdout=fo, stderr=subprocess.PIPE, env=env, check=True)
        wall = time.perf_counter() - t0
        line = [x for x in p.stderr.decode().splitlines() if "stage 1" in x][-1]
        f = [float(x) for x in re.findall(r"(\d+\.\d+)", line)]
        # open | read, tokenize, group, wait | device, format+write; the command's own clock from "index and taxonomy resident" to "last read written"
        stream = float(re.search(r"stream ([0-9.]+) s", p.stderr.decode()).group(1))
        res["runs"][name].append({"wall_s": round(wall, 3), "open_s": f[0], "read": f[1], "tokenize": f[2], "group": f[3], "wait": f[4], "device": f[5], "format_write": f[6],
                                  "stream_s": round(stream, 3), "M_pairs_per_s_steady": round(n / stream / 1e6, 1)})
        with open(out, "rb") as fh:
            digests.add(hashlib.sha256(fh.read()).hexdigest())
for name in res["runs"]:
    v = sorted(x["stream_s"] for x in res["runs"][name])
    res[name + "_stream_s_min_median"] = [v[0], v[len(v) // 2]]
    g = sorted(x["group"] for x in res["runs"][name])
    res[name + "_group_s_median"] = g[len(g) // 2]
res["outputs_identical"] = len(digests) == 1
print(json.dumps(res))
