#!/usr/bin/env python3
"""Phase times (UMGAP_TIMING=1) of `umgap prot2kmer2lca -o` alone over a file: python tools/stage_timing.py [n_pairs]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import end_to_end as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
work = "/tmp/umgap_e2e"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs(work, n, 250_000)
env = dict(os.environ, UMGAP_TIMING="1")
for _ in range(2):
    t0 = time.perf_counter()
    with open(fasta, "rb") as fi, open(os.path.join(work, "out.lca"), "wb") as fo:
        p = subprocess.run([E.UMGAP, "prot2kmer2lca", "-o", idx], stdin=fi, stdout=fo, stderr=subprocess.PIPE, env=env, check=True)
    dt = time.perf_counter() - t0
print("prot2kmer2lca -o: %d pairs in %.3f s = %.2f M pairs/s; %.0f MB in, %.0f MB out" % (n, dt, n / dt / 1e6, os.path.getsize(fasta) / 1e6, os.path.getsize(os.path.join(work, "out.lca")) / 1e6))
print(p.stderr.decode().strip())
